// A4 / A5: sparse x sparse product (replaces `_csr_csr_count_nnz` + `_dot_csr_csr` and
// `_dot_coo_coo`, sparse/numba_backend/_common.py:543-570,639-717,907-976).
//
// The reference is Gustavson's row-wise algorithm with a dense accumulator `sums[n_col]` and an
// intrusive linked list per row — one thread, n_col-sized scratch per row, unsorted output rows.
// Here the same products are formed by expand -> stable sort -> compress (ESC):
//   1. spamd_spgemm_count   cnt[p] = nnz of B row a_indices[p]             (one pass over A)
//   2. exclusive scan       product offsets, total P                      (prims.hip)
//   3. spamd_spgemm_expand  key[t] = row(p)*n_col + b_col, val[t] = a*b   (P products, coalesced
//                           over t; the owning A element is found by binary search in offsets)
//   4. stable radix sort by key, head flags, segment_reduce(add)          (prims.hip / ewise.hip)
// Products of one output element stay in generation order (A's k ascending, then B's storage
// order) through the stable sort, and the run is summed left to right, so the floating-point
// result is bit-identical to the reference's `sums[k] += av * bv` sequence; the output rows
// come out SORTED by column (the reference's are in reverse discovery order — compared after a
// canonical sort, Appendix C.2).  Memory is bounded by chunking rows on the host (_kernels.py).
#include "common.h"

namespace spamd {

#define GRID_STRIDE(i, n)                                                          \
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n);        \
       i += (int64_t)gridDim.x * blockDim.x)

static inline unsigned grid_for(int64_t n) {
  int64_t b = ceil_div(n, 256);
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (unsigned)b;
}

template <typename I>
__global__ void __launch_bounds__(256) spgemm_count_kernel(const I* __restrict__ a_idx, int64_t p0, int64_t np,
                                                           const I* __restrict__ b_ptr, int64_t* __restrict__ cnt) {
  GRID_STRIDE(i, np) {
    const int64_t c = (int64_t)a_idx[p0 + i];
    cnt[i] = (int64_t)b_ptr[c + 1] - (int64_t)b_ptr[c];
  }
}

// rows: a_rows[p] (row id of A element p) — for CSR operands produced by spamd_csr_to_keys / ncolA
template <typename T, typename I>
__global__ void __launch_bounds__(256) spgemm_expand_kernel(
    const T* __restrict__ a_data, const I* __restrict__ a_idx, const int64_t* __restrict__ a_rows, int64_t p0,
    int64_t np, const T* __restrict__ b_data, const I* __restrict__ b_idx, const I* __restrict__ b_ptr,
    const int64_t* __restrict__ offs, int64_t P, int64_t n_col, int64_t* __restrict__ keys, T* __restrict__ vals) {
#pragma clang fp contract(off)
  GRID_STRIDE(t, P) {
    int64_t lo = 0, hi = np - 1;  // last p with offs[p] <= t
    while (lo < hi) {
      const int64_t mid = (lo + hi + 1) >> 1;
      if (offs[mid] <= t) lo = mid; else hi = mid - 1;
    }
    const int64_t p = p0 + lo;
    const int64_t q = (int64_t)b_ptr[(int64_t)a_idx[p]] + (t - offs[lo]);
    keys[t] = a_rows[p] * n_col + (int64_t)b_idx[q];
    vals[t] = a_data[p] * b_data[q];
  }
}

}  // namespace spamd

using namespace spamd;

extern "C" int spamd_spgemm_count(int idx_dtype, int64_t p0, int64_t np, const void* a_indices, const void* b_indptr,
                                  int64_t* cnt, void* stream) {
  if (np < 0 || p0 < 0) return SPAMD_EINVAL;
  if (np == 0) return 0;
  SPAMD_DISPATCH_IDX(idx_dtype, I, hipLaunchKernelGGL(spgemm_count_kernel<I>, dim3(grid_for(np)), dim3(256), 0,
                                                      (hipStream_t)stream, (const I*)a_indices, p0, np,
                                                      (const I*)b_indptr, cnt))
  return launch_status();
}

extern "C" int spamd_spgemm_expand(int val_dtype, int idx_dtype, int64_t p0, int64_t np, const void* a_data,
                                   const void* a_indices, const int64_t* a_rows, const void* b_data,
                                   const void* b_indices, const void* b_indptr, const int64_t* offsets, int64_t P,
                                   int64_t n_col, int64_t* keys, void* vals, void* stream) {
  if (np < 0 || P < 0 || n_col < 0) return SPAMD_EINVAL;
  if (P == 0 || np == 0) return 0;
  SPAMD_DISPATCH_VAL(val_dtype, T, {
    SPAMD_DISPATCH_IDX(idx_dtype, I, hipLaunchKernelGGL((spgemm_expand_kernel<T, I>), dim3(grid_for(P)), dim3(256), 0,
                                                        (hipStream_t)stream, (const T*)a_data, (const I*)a_indices,
                                                        a_rows, p0, np, (const T*)b_data, (const I*)b_indices,
                                                        (const I*)b_indptr, offsets, P, n_col, keys, (T*)vals))
  })
  return launch_status();
}
