// Rows far longer than the rest (the hub of a graph: 10^5 .. 10^6 stored elements in one row of a matrix whose mean row holds
// 10) are walked by ONE wave in every form of CSR x dense here (reference `_dot_csr_ndarray`, sparse/numba_backend/
// _common.py:720-755, walks them on one core too): 117 ms for a row of 10^6 elements in a product that takes 0.64 ms
// without it (tools/r06/skew_probe.py, round 6).  `_dot.py` multiplies such an operand in two parts: the matrix without its
// hot rows, and the hot rows cut into pieces that are rows of their own (`_hot_row_split`) - both through the kernels that
// exist -, and this kernel adds a hot row's pieces into its row of the result, in piece order (reproducible).
#include "common.h"

namespace spamd {

// out[rows[h], :] = sum of part[vfirst[h] .. vfirst[h + 1], :]; a workgroup per (hot row, 256-column slab)
template <typename T>
__global__ void __launch_bounds__(256) hot_rows_combine_kernel(int64_t n_cols, const T* __restrict__ part, int64_t ld_part,
                                                                const int64_t* __restrict__ vfirst, const int64_t* __restrict__ rows,
                                                                T* __restrict__ out, int64_t ld_out, int64_t slabs) {
  const int64_t h = (int64_t)blockIdx.x / slabs, c = ((int64_t)blockIdx.x % slabs) * 256 + threadIdx.x;
  if (c >= n_cols) return;
  const int64_t v0 = vfirst[h], v1 = vfirst[h + 1];
  T acc = T(0);
#pragma unroll 8
  for (int64_t v = v0; v < v1; ++v) acc += part[v * ld_part + c];      // (independent loads: eight in flight)
  out[rows[h] * ld_out + c] = acc;
}

}  // namespace spamd

extern "C" int spamd_hot_rows_combine(int val_dtype, int64_t n_hot, int64_t n_cols, const void* part, int64_t ld_part,
                                      const int64_t* vfirst, const int64_t* rows, void* out, int64_t ld_out, void* stream) {
  using namespace spamd;
  if (n_hot < 0 || n_cols < 0 || ld_part < n_cols || ld_out < n_cols) return SPAMD_EINVAL;
  if (n_hot == 0 || n_cols == 0) return 0;
  const int64_t slabs = ceil_div(n_cols, (int64_t)256);
  if (n_hot * slabs >= ((int64_t)1 << 31)) return SPAMD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)(n_hot * slabs));
#define HR_CASE(CODE, T)                                                                                                        \
  case CODE:                                                                                                                    \
    hipLaunchKernelGGL(hot_rows_combine_kernel<T>, grid, dim3(256), 0, s, n_cols, (const T*)part, ld_part, vfirst, rows, (T*)out, \
                       ld_out, slabs);                                                                                          \
    break;
  switch (val_dtype) {
    HR_CASE(SPAMD_F32, float)
    HR_CASE(SPAMD_F64, double)
    HR_CASE(SPAMD_I32, int32_t)
    HR_CASE(SPAMD_I64, int64_t)
    default: return SPAMD_ETYPE;
  }
#undef HR_CASE
  return launch_status();
}
