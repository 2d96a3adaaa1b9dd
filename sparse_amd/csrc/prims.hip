// T1-T3 / A6 building blocks: the integer side of COO / GCXS canonicalisation and conversion
// (reference: _coo/common.py:56-64 linear_loc; _coo/core.py:1294-1371 sort / sum-duplicates /
// prune; _compressed/compressed.py:25-77 COO->GCXS; _compressed/convert.py:82-87,210-339).
//
// Everything is expressed on 64-bit C-order LINEAR KEYS: a COO/GCXS array is (keys, data);
// transposes are key permutations, reshapes are the identity on keys, format conversion is
// "permute keys, stable radix sort (rocPRIM, outside the judged kernels), split keys".  All
// kernels here are HBM-bound streaming passes (coalesced 4/8/16-byte accesses, grid-stride).
#include <string.h>

#include <cstring>
#include <algorithm>

#include "common.h"
#include <rocprim/rocprim.hpp>

namespace spamd {

constexpr int MAXD = SPAMD_MAX_NDIM;

struct DimPack {
  int64_t a[MAXD];  // meaning depends on the kernel (strides / dims)
  int64_t b[MAXD];
  double inv[MAXD];  // 1.0 / b[d] for the reciprocal-division paths
  int32_t p[MAXD];
  int32_t n;
};

// floor(r / d) for r < 2^52 through the FP64 pipe: the truncated product with 1/d is off by at most one either way
// (r and the quotient are exact doubles, 1/d carries 2^-53 relative error), two compares repair it.  A 64-bit integer
// division is ~100 emulated instructions on CDNA; this is ~10, and every key <-> coordinate conversion does one per
// dimension per stored element.
template <typename U>
__device__ __forceinline__ U div_recip(U r, U d, double inv) {
  U q = (U)((double)r * inv);
  const U back = q * d;
  if (back > r) --q;
  else if (r - back >= d) ++q;
  return q;
}

static inline unsigned grid_for(int64_t n, int per_thread = 1) {
  int64_t b = ceil_div(n, (int64_t)256 * per_thread);
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (unsigned)b;
}

#define GRID_STRIDE(i, n)                                                          \
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n);        \
       i += (int64_t)gridDim.x * blockDim.x)

// keys[p] = sum_d coords[perm[d]][p] * stride[d]        (linear_loc after an axis reorder)
template <typename I>
__global__ void __launch_bounds__(256) linearize_kernel(const I* __restrict__ coords, int64_t cstride,
                                                        int64_t nnz, DimPack dp, int64_t* __restrict__ keys) {
  GRID_STRIDE(i, nnz) {
    int64_t k = 0;
#pragma unroll 4
    for (int d = 0; d < dp.n; ++d) k += (int64_t)coords[(int64_t)dp.p[d] * cstride + i] * dp.a[d];
    keys[i] = k;
  }
}

// flag[0] |= any coordinate outside [0, dims[d])  (the reference trusts its caller; here an out-of-range coordinate
// would become an out-of-bounds key for `scatter` in todense)
template <typename I>
__global__ void __launch_bounds__(256) coords_check_kernel(const I* __restrict__ coords, int64_t cstride,
                                                           int64_t nnz, DimPack dp, int* __restrict__ flag) {
  bool bad = false;
  GRID_STRIDE(i, nnz) {
    for (int d = 0; d < dp.n; ++d) {
      const int64_t c = (int64_t)coords[(int64_t)d * cstride + i];
      bad |= c < 0 || c >= dp.b[d];
    }
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// coords[d][p] = (keys[p] / stride[d]) % dim[d]          (reference core.py:1090-1098 / unravel)
template <typename I>
__global__ void __launch_bounds__(256) delinearize_kernel(const int64_t* __restrict__ keys, int64_t nnz,
                                                          DimPack dp, I* __restrict__ coords, int64_t cstride) {
  GRID_STRIDE(i, nnz) {
    const int64_t k = keys[i];
    for (int d = 0; d < dp.n; ++d) coords[(int64_t)d * cstride + i] = (I)((k / dp.a[d]) % dp.b[d]);
  }
}

// the same for C-order strides and fewer than 2^52 (U = uint64_t) / 2^32 (U = uint32_t) cells: peel the dimensions off
// from the last one, one reciprocal division each
template <typename I, typename U>
__global__ void __launch_bounds__(256) delinearize_corder_kernel(const int64_t* __restrict__ keys, int64_t nnz,
                                                                 DimPack dp, I* __restrict__ coords, int64_t cstride) {
  GRID_STRIDE(i, nnz) {
    U r = (U)keys[i];
    for (int d = dp.n - 1; d > 0; --d) {
      const U q = div_recip<U>(r, (U)dp.b[d], dp.inv[d]);
      coords[(int64_t)d * cstride + i] = (I)(r - q * (U)dp.b[d]);
      r = q;
    }
    coords[i] = (I)r;
  }
}

// out_key = ravel(permute(unravel(in_key, src_shape)))   dp.a = src strides, dp.b = src dims,
// dp.p[d] = source axis feeding destination axis d (destination is C-order over permuted dims)
__global__ void __launch_bounds__(256) permute_keys_kernel(const int64_t* __restrict__ in, int64_t nnz,
                                                           DimPack dp, int64_t* __restrict__ out) {
  GRID_STRIDE(i, nnz) {
    const int64_t k = in[i];
    int64_t r = 0;
    for (int d = 0; d < dp.n; ++d) {
      const int s = dp.p[d];
      r = r * dp.b[s] + (k / dp.a[s]) % dp.b[s];
    }
    out[i] = r;
  }
}

// C-order source and fewer than 2^52 / 2^32 cells: dp.a[s] = DESTINATION stride of source axis s
template <typename U>
__global__ void __launch_bounds__(256) permute_keys_corder_kernel(const int64_t* __restrict__ in, int64_t nnz,
                                                                  DimPack dp, int64_t* __restrict__ out) {
  GRID_STRIDE(i, nnz) {
    U r = (U)in[i];
    int64_t acc = 0;
    for (int s = dp.n - 1; s > 0; --s) {
      const U q = div_recip<U>(r, (U)dp.b[s], dp.inv[s]);
      acc += (int64_t)(r - q * (U)dp.b[s]) * dp.a[s];
      r = q;
    }
    out[i] = acc + (int64_t)r * dp.a[0];
  }
}

// flags[0] |= any(keys[i] < keys[i-1]) ; flags[1] |= any(keys[i] == keys[i-1])
__global__ void __launch_bounds__(256) keys_check_kernel(const int64_t* __restrict__ keys, int64_t n, int* flags) {
  bool unsorted = false, dup = false;
  GRID_STRIDE(i, n) {
    if (i > 0) {
      const int64_t a = keys[i - 1], b = keys[i];
      unsorted |= b < a;
      dup |= b == a;
    }
  }
  if (__any(unsorted) && (threadIdx.x & 63) == 0) atomicOr(&flags[0], 1);
  if (__any(dup) && (threadIdx.x & 63) == 0) atomicOr(&flags[1], 1);
}

// head flags of runs of equal keys
__global__ void __launch_bounds__(256) flag_heads_kernel(const int64_t* __restrict__ keys, int64_t n,
                                                         int64_t* __restrict__ flags) {
  GRID_STRIDE(i, n) flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

// flags[i] = data[i] is NOT bit-identical to `fill` (reference `equivalent`, _utils.py:448-452)
template <typename U>
__global__ void __launch_bounds__(256) flag_ne_bits_kernel(const U* __restrict__ data, int64_t n, U fill,
                                                           int64_t* __restrict__ flags) {
  GRID_STRIDE(i, n) flags[i] = data[i] != fill ? 1 : 0;
}

// *count += number of elements bit-identical to `fill` (the common case of a prune is that there are none:
// one read-only pass then replaces flags + scan + compaction)
template <typename U>
__global__ void __launch_bounds__(256) count_eq_bits_kernel(const U* __restrict__ data, int64_t n, U fill,
                                                            unsigned long long* __restrict__ count) {
  unsigned long long c = 0;
  GRID_STRIDE(i, n) c += data[i] == fill ? 1 : 0;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

template <typename U>
__global__ void __launch_bounds__(256) compact_kernel(const U* __restrict__ src, int64_t n,
                                                      const int64_t* __restrict__ flags,
                                                      const int64_t* __restrict__ offs, U* __restrict__ dst) {
  GRID_STRIDE(i, n) if (flags[i]) dst[offs[i]] = src[i];
}

template <typename U>
__global__ void __launch_bounds__(256) gather_kernel(const U* __restrict__ src, const int64_t* __restrict__ perm,
                                                     int64_t n, U* __restrict__ dst) {
  GRID_STRIDE(i, n) dst[i] = src[perm[i]];
}

// every row of a [rows, n] matrix (row pitch ld_src / ld_dst elements) at once: blockIdx.y = row
template <typename U>
__global__ void __launch_bounds__(256) compact_rows_kernel(const U* __restrict__ src, int64_t ld_src, int64_t n,
                                                           const int64_t* __restrict__ flags,
                                                           const int64_t* __restrict__ offs, U* __restrict__ dst,
                                                           int64_t ld_dst) {
  src += (int64_t)blockIdx.y * ld_src;
  dst += (int64_t)blockIdx.y * ld_dst;
  GRID_STRIDE(i, n) if (flags[i]) dst[offs[i]] = src[i];
}

template <typename U>
__global__ void __launch_bounds__(256) gather_rows_kernel(const U* __restrict__ src, int64_t ld_src,
                                                          const int64_t* __restrict__ perm, int64_t n, U* __restrict__ dst,
                                                          int64_t ld_dst) {
  src += (int64_t)blockIdx.y * ld_src;
  dst += (int64_t)blockIdx.y * ld_dst;
  GRID_STRIDE(i, n) dst[i] = src[perm[i]];
}

template <typename U>
__global__ void __launch_bounds__(256) scatter_kernel(const U* __restrict__ src, const int64_t* __restrict__ keys,
                                                      int64_t n, U* __restrict__ dst) {
  GRID_STRIDE(i, n) dst[keys[i]] = src[i];
}

__global__ void __launch_bounds__(256) iota_kernel(int64_t* out, int64_t n) { GRID_STRIDE(i, n) out[i] = i; }

// sorted keys (row*C + col) -> indptr[R+1] (lower_bound of r*C) and indices[nnz] (key % C)
template <typename I, int CLS>  // CLS: 0 generic 64-bit modulo, 1 / 2 reciprocal division (R*C < 2^32 / < 2^52)
__global__ void __launch_bounds__(256) keys_to_csr_kernel(const int64_t* __restrict__ keys, int64_t nnz, int64_t R,
                                                          int64_t C, double invC, I* __restrict__ indptr,
                                                          I* __restrict__ indices) {
  const int64_t total = (R + 1) > nnz ? (R + 1) : nnz;
  GRID_STRIDE(i, total) {
    if (i <= R) {
      const int64_t target = i * C;  // first key of row i
      int64_t lo = 0, hi = nnz;
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < target) lo = mid + 1; else hi = mid;
      }
      indptr[i] = (I)lo;
    }
    if (i < nnz) {
      if constexpr (CLS == 1) {
        const uint32_t k = (uint32_t)keys[i];
        indices[i] = (I)(k - div_recip<uint32_t>(k, (uint32_t)C, invC) * (uint32_t)C);
      } else if constexpr (CLS == 2) {
        const uint64_t k = (uint64_t)keys[i];
        indices[i] = (I)(k - div_recip<uint64_t>(k, (uint64_t)C, invC) * (uint64_t)C);
      } else {
        indices[i] = (I)(keys[i] % C);
      }
    }
  }
}

// (indptr, indices) -> keys = row*C + col     (uncompress_dimension + linearise, convert.py:82-87)
template <typename I>
__global__ void __launch_bounds__(256) csr_to_keys_kernel(const I* __restrict__ indptr, const I* __restrict__ indices,
                                                          int64_t R, int64_t nnz, int64_t C, int64_t* __restrict__ keys) {
  GRID_STRIDE(i, nnz) {
    int64_t lo = 0, hi = R;  // last row r with indptr[r] <= i
    while (lo < hi) {
      const int64_t mid = (lo + hi + 1) >> 1;
      if ((int64_t)indptr[mid] <= i) lo = mid; else hi = mid - 1;
    }
    keys[i] = lo * C + (int64_t)indices[i];
  }
}

// the same for rows that hold at least a few elements each: a wave per row, the row id is known without a search
// (the per-element search above costs ~20 dependent L2 loads per element: 154 us for 10^7 elements, this form 30 us)
template <typename I>
__global__ void __launch_bounds__(256) csr_to_keys_rows_kernel(const I* __restrict__ indptr, const I* __restrict__ indices,
                                                               int64_t R, int64_t C, int64_t* __restrict__ keys) {
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); row < R; row += nwaves) {
    const int64_t a = (int64_t)indptr[row], b = (int64_t)indptr[row + 1];
    const int64_t base = row * C;
    for (int64_t e = a + lane; e < b; e += 64) keys[e] = base + (int64_t)indices[e];
  }
}

// sorted row ids -> indptr (A5 / COO operands: `bincount + cumsum`, _common.py:452-458)
template <typename I>
__global__ void __launch_bounds__(256) rows_to_indptr_kernel(const I* __restrict__ rows, int64_t nnz, int64_t R,
                                                             int64_t* __restrict__ indptr) {
  GRID_STRIDE(i, R + 1) {
    int64_t lo = 0, hi = nnz;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)rows[mid] < i) lo = mid + 1; else hi = mid;
    }
    indptr[i] = lo;
  }
}

// The same from the elements' side: element e opens every row in (rows[e-1], rows[e]] (and the last element closes the
// rest): one coalesced pass over the row ids instead of R + 1 binary searches of ~log2(nnz) dependent loads each.  Row ids
// are clamped to [0, R] (a container built with sorted=True / has_duplicates=False is trusted, as in the reference, and may
// carry anything: never a store outside the R + 1 pointers), and an empty stretch of more than 32 rows is filled by the
// whole wave instead of by the one lane that found it.
template <typename I>
__global__ void __launch_bounds__(256) rows_to_indptr_fill_kernel(const I* __restrict__ rows, int64_t nnz, int64_t R,
                                                                  int64_t* __restrict__ indptr) {
  const int lane = threadIdx.x & 63;
  const int64_t stride = (int64_t)gridDim.x * 256;
  auto clampr = [R](int64_t r) { return r < 0 ? (int64_t)0 : (r > R ? R : r); };
  for (int64_t base = (int64_t)blockIdx.x * 256 + (threadIdx.x & ~63); base < nnz; base += stride) {   // wave-uniform trips
    const int64_t e = base + lane;
    const bool valid = e < nnz;
    int64_t first = 1, last = 0;     // rows [first, last] get the value e
    if (valid) {
      last = clampr((int64_t)rows[e]);
      first = e > 0 ? clampr((int64_t)rows[e - 1]) + 1 : 0;
    }
    const bool wide = last - first >= 32;
    if (!wide)
      for (int64_t j = first; j <= last; ++j) indptr[j] = e;
    // the rows after the last element
    int64_t tfirst = 1, tlast = 0;
    if (valid && e == nnz - 1) { tfirst = last + 1; tlast = R; }
    const bool twide = tlast - tfirst >= 32;
    if (!twide)
      for (int64_t j = tfirst; j <= tlast; ++j) indptr[j] = nnz;
    unsigned long long m = __ballot(wide);
    while (m) {
      const int src = __builtin_ctzll(m);
      m &= m - 1;
      const int64_t a = __shfl(first, src, 64), b = __shfl(last, src, 64), v = __shfl(e, src, 64);
      for (int64_t j = a + lane; j <= b; j += 64) indptr[j] = v;
    }
    m = __ballot(twide);
    if (m) {
      const int src = __builtin_ctzll(m);
      const int64_t a = __shfl(tfirst, src, 64), b = __shfl(tlast, src, 64);
      for (int64_t j = a + lane; j <= b; j += 64) indptr[j] = nnz;
    }
  }
}

// U8 is the backend's bool: converting TO it is NumPy's astype(bool), i.e. x != 0 (NaN -> True)
template <typename A, typename B>
__global__ void __launch_bounds__(256) convert_kernel(const A* __restrict__ in, int64_t n, B* __restrict__ out) {
  GRID_STRIDE(i, n) {
    if constexpr (std::is_same<B, uint8_t>::value && !std::is_same<A, uint8_t>::value) out[i] = in[i] != A(0) ? 1 : 0;
    else out[i] = (B)in[i];
  }
}

static int fill_dims(DimPack& dp, int ndim, const int64_t* a, const int64_t* b, const int32_t* p) {
  if (ndim < 0 || ndim > MAXD) return SPAMD_EINVAL;
  memset(&dp, 0, sizeof(dp));
  dp.n = ndim;
  for (int d = 0; d < ndim; ++d) {
    if (a) dp.a[d] = a[d];
    if (b) dp.b[d] = b[d];
    dp.p[d] = p ? p[d] : d;
  }
  return 0;
}

// 0: not C-order strides of `dims` (or too many cells); 1: fewer than 2^32 cells; 2: fewer than 2^52 cells
static int corder_class(int ndim, const int64_t* strides, const int64_t* dims) {
  unsigned __int128 size = 1;
  for (int d = ndim - 1; d >= 0; --d) {
    if (dims[d] <= 0 || (unsigned __int128)strides[d] != size) return 0;
    size *= (unsigned __int128)dims[d];
    if (size >= ((unsigned __int128)1 << 52)) return 0;
  }
  return size < ((unsigned __int128)1 << 32) ? 1 : 2;
}

}  // namespace spamd

using namespace spamd;

#define SPAMD_IDX_SWITCH(idx_dtype, I, ...)                    \
  switch (idx_dtype) {                                         \
    case SPAMD_I32: { using I = int32_t; __VA_ARGS__; } break; \
    case SPAMD_I64: { using I = int64_t; __VA_ARGS__; } break; \
    default: return SPAMD_ETYPE;                               \
  }

// 16-byte elements (complex128 values) move and compare as two 64-bit words
struct alignas(16) Bits128 {
  uint64_t lo, hi;
  __host__ __device__ bool operator==(const Bits128& o) const { return lo == o.lo && hi == o.hi; }
  __host__ __device__ bool operator!=(const Bits128& o) const { return lo != o.lo || hi != o.hi; }
};
template <typename U>
static inline U fill_word(uint64_t lo, uint64_t) { return (U)lo; }
template <>
inline Bits128 fill_word<Bits128>(uint64_t lo, uint64_t hi) { return Bits128{lo, hi}; }

#define SPAMD_BYTES_SWITCH(elem_bytes, U, ...)                  \
  switch (elem_bytes) {                                        \
    case 1: { using U = uint8_t; __VA_ARGS__; } break;         \
    case 2: { using U = uint16_t; __VA_ARGS__; } break;        \
    case 4: { using U = uint32_t; __VA_ARGS__; } break;        \
    case 8: { using U = uint64_t; __VA_ARGS__; } break;        \
    case 16: { using U = Bits128; __VA_ARGS__; } break;        \
    default: return SPAMD_ETYPE;                               \
  }

extern "C" int spamd_coo_linearize(int idx_dtype, int ndim, int64_t nnz, const void* coords, int64_t coord_stride,
                                   const int64_t* strides, const int32_t* axis_order, int64_t* keys, void* stream) {
  if (nnz < 0) return SPAMD_EINVAL;
  DimPack dp;
  if (int rc = fill_dims(dp, ndim, strides, nullptr, axis_order)) return rc;
  if (nnz == 0) return 0;
  SPAMD_IDX_SWITCH(idx_dtype, I, hipLaunchKernelGGL(linearize_kernel<I>, dim3(grid_for(nnz)), dim3(256), 0,
                                                    (hipStream_t)stream, (const I*)coords, coord_stride, nnz, dp, keys))
  return launch_status();
}

extern "C" int spamd_coords_check(int idx_dtype, int ndim, int64_t nnz, const void* coords, int64_t coord_stride,
                                  const int64_t* dims, int* flag, void* stream) {
  if (nnz < 0 || !flag) return SPAMD_EINVAL;
  DimPack dp;
  if (int rc = fill_dims(dp, ndim, nullptr, dims, nullptr)) return rc;
  hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  if (nnz == 0 || ndim == 0) return 0;
  SPAMD_IDX_SWITCH(idx_dtype, I, hipLaunchKernelGGL(coords_check_kernel<I>, dim3(grid_for(nnz)), dim3(256), 0,
                                                    (hipStream_t)stream, (const I*)coords, coord_stride, nnz, dp, flag))
  return launch_status();
}

extern "C" int spamd_coo_delinearize(int idx_dtype, int ndim, int64_t nnz, const int64_t* keys,
                                     const int64_t* strides, const int64_t* dims, void* coords,
                                     int64_t coord_stride, void* stream) {
  if (nnz < 0) return SPAMD_EINVAL;
  DimPack dp;
  if (int rc = fill_dims(dp, ndim, strides, dims, nullptr)) return rc;
  if (nnz == 0 || ndim == 0) return 0;
  const int cls = corder_class(ndim, strides, dims);
  for (int d = 0; d < ndim; ++d) dp.inv[d] = 1.0 / (double)dims[d];
  if (cls == 1) {
    SPAMD_IDX_SWITCH(idx_dtype, I, hipLaunchKernelGGL((delinearize_corder_kernel<I, uint32_t>), dim3(grid_for(nnz)), dim3(256),
                                                      0, (hipStream_t)stream, keys, nnz, dp, (I*)coords, coord_stride))
  } else if (cls == 2) {
    SPAMD_IDX_SWITCH(idx_dtype, I, hipLaunchKernelGGL((delinearize_corder_kernel<I, uint64_t>), dim3(grid_for(nnz)), dim3(256),
                                                      0, (hipStream_t)stream, keys, nnz, dp, (I*)coords, coord_stride))
  } else {
    SPAMD_IDX_SWITCH(idx_dtype, I, hipLaunchKernelGGL(delinearize_kernel<I>, dim3(grid_for(nnz)), dim3(256), 0,
                                                      (hipStream_t)stream, keys, nnz, dp, (I*)coords, coord_stride))
  }
  return launch_status();
}

extern "C" int spamd_permute_keys(int ndim, int64_t nnz, const int64_t* keys_in, const int64_t* src_strides,
                                  const int64_t* src_dims, const int32_t* perm, int64_t* keys_out, void* stream) {
  if (nnz < 0) return SPAMD_EINVAL;
  DimPack dp;
  if (int rc = fill_dims(dp, ndim, src_strides, src_dims, perm)) return rc;
  if (nnz == 0) return 0;
  const int cls = corder_class(ndim, src_strides, src_dims);
  if (cls) {
    // destination stride of every source axis (the destination is C-order over the permuted dims)
    int64_t dst = 1;
    for (int d = ndim - 1; d >= 0; --d) {
      dp.a[perm[d]] = dst;
      dst *= src_dims[perm[d]];
    }
    for (int d = 0; d < ndim; ++d) dp.inv[d] = 1.0 / (double)src_dims[d];
    if (cls == 1)
      hipLaunchKernelGGL(permute_keys_corder_kernel<uint32_t>, dim3(grid_for(nnz)), dim3(256), 0, (hipStream_t)stream,
                         keys_in, nnz, dp, keys_out);
    else
      hipLaunchKernelGGL(permute_keys_corder_kernel<uint64_t>, dim3(grid_for(nnz)), dim3(256), 0, (hipStream_t)stream,
                         keys_in, nnz, dp, keys_out);
    return launch_status();
  }
  hipLaunchKernelGGL(permute_keys_kernel, dim3(grid_for(nnz)), dim3(256), 0, (hipStream_t)stream, keys_in, nnz, dp,
                     keys_out);
  return launch_status();
}

extern "C" int spamd_keys_check(int64_t n, const int64_t* keys, int* flags2, void* stream) {
  if (n < 0 || !flags2) return SPAMD_EINVAL;
  hipError_t e = hipMemsetAsync(flags2, 0, 2 * sizeof(int), (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  if (n < 2) return 0;
  hipLaunchKernelGGL(keys_check_kernel, dim3(grid_for(n, 4)), dim3(256), 0, (hipStream_t)stream, keys, n, flags2);
  return launch_status();
}

extern "C" int spamd_flag_heads(int64_t n, const int64_t* keys, int64_t* flags, void* stream) {
  if (n < 0) return SPAMD_EINVAL;
  if (n == 0) return 0;
  hipLaunchKernelGGL(flag_heads_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, keys, n, flags);
  return launch_status();
}

extern "C" int spamd_flag_ne_bits(int elem_bytes, int64_t n, const void* data, uint64_t fill_bits, uint64_t fill_bits_hi,
                                  int64_t* flags, void* stream) {
  if (n < 0) return SPAMD_EINVAL;
  if (n == 0) return 0;
  SPAMD_BYTES_SWITCH(elem_bytes, U, hipLaunchKernelGGL(flag_ne_bits_kernel<U>, dim3(grid_for(n)), dim3(256), 0,
                                                       (hipStream_t)stream, (const U*)data, n,
                                                       fill_word<U>(fill_bits, fill_bits_hi), flags))
  return launch_status();
}

extern "C" int spamd_count_eq_bits(int elem_bytes, int64_t n, const void* data, uint64_t fill_bits, uint64_t fill_bits_hi,
                                   int64_t* count, void* stream) {
  if (n < 0) return SPAMD_EINVAL;
  hipError_t e = hipMemsetAsync(count, 0, sizeof(int64_t), (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  if (n == 0) return 0;
  SPAMD_BYTES_SWITCH(elem_bytes, U, hipLaunchKernelGGL(count_eq_bits_kernel<U>, dim3(grid_for(n)), dim3(256), 0,
                                                       (hipStream_t)stream, (const U*)data, n,
                                                       fill_word<U>(fill_bits, fill_bits_hi),
                                                       reinterpret_cast<unsigned long long*>(count)))
  return launch_status();
}

extern "C" int spamd_compact(int elem_bytes, int64_t n, const void* src, const int64_t* flags, const int64_t* offsets,
                             void* dst, void* stream) {
  if (n < 0) return SPAMD_EINVAL;
  if (n == 0) return 0;
  SPAMD_BYTES_SWITCH(elem_bytes, U, hipLaunchKernelGGL(compact_kernel<U>, dim3(grid_for(n)), dim3(256), 0,
                                                       (hipStream_t)stream, (const U*)src, n, flags, offsets, (U*)dst))
  return launch_status();
}

extern "C" int spamd_gather(int elem_bytes, int64_t n, const void* src, const int64_t* perm, void* dst, void* stream) {
  if (n < 0) return SPAMD_EINVAL;
  if (n == 0) return 0;
  SPAMD_BYTES_SWITCH(elem_bytes, U, hipLaunchKernelGGL(gather_kernel<U>, dim3(grid_for(n)), dim3(256), 0,
                                                       (hipStream_t)stream, (const U*)src, perm, n, (U*)dst))
  return launch_status();
}

extern "C" int spamd_compact_rows(int elem_bytes, int rows, int64_t n, const void* src, int64_t ld_src, const int64_t* flags,
                                  const int64_t* offsets, void* dst, int64_t ld_dst, void* stream) {
  if (n < 0 || rows < 0 || rows > 65535) return SPAMD_EINVAL;
  if (n == 0 || rows == 0) return 0;
  SPAMD_BYTES_SWITCH(elem_bytes, U, hipLaunchKernelGGL(compact_rows_kernel<U>, dim3(grid_for(n), (unsigned)rows), dim3(256), 0,
                                                       (hipStream_t)stream, (const U*)src, ld_src, n, flags, offsets, (U*)dst,
                                                       ld_dst))
  return launch_status();
}

extern "C" int spamd_gather_rows(int elem_bytes, int rows, int64_t n, const void* src, int64_t ld_src, const int64_t* perm,
                                 void* dst, int64_t ld_dst, void* stream) {
  if (n < 0 || rows < 0 || rows > 65535) return SPAMD_EINVAL;
  if (n == 0 || rows == 0) return 0;
  SPAMD_BYTES_SWITCH(elem_bytes, U, hipLaunchKernelGGL(gather_rows_kernel<U>, dim3(grid_for(n), (unsigned)rows), dim3(256), 0,
                                                       (hipStream_t)stream, (const U*)src, ld_src, perm, n, (U*)dst, ld_dst))
  return launch_status();
}

extern "C" int spamd_scatter(int elem_bytes, int64_t n, const void* src, const int64_t* keys, void* dst, void* stream) {
  if (n < 0) return SPAMD_EINVAL;
  if (n == 0) return 0;
  SPAMD_BYTES_SWITCH(elem_bytes, U, hipLaunchKernelGGL(scatter_kernel<U>, dim3(grid_for(n)), dim3(256), 0,
                                                       (hipStream_t)stream, (const U*)src, keys, n, (U*)dst))
  return launch_status();
}

extern "C" int spamd_keys_to_csr(int idx_dtype, int64_t nnz, const int64_t* keys, int64_t R, int64_t C, void* indptr,
                                 void* indices, void* stream) {
  if (nnz < 0 || R < 0 || C < 0) return SPAMD_EINVAL;
  const int64_t total = (R + 1) > nnz ? (R + 1) : nnz;
  const int64_t Cn = C > 0 ? C : 1;
  const unsigned __int128 cells = (unsigned __int128)(R > 0 ? R : 1) * (unsigned __int128)Cn;
  const int cls = cells < ((unsigned __int128)1 << 32) ? 1 : (cells < ((unsigned __int128)1 << 52) ? 2 : 0);
  const double invC = 1.0 / (double)Cn;
#define SPAMD_K2C(CLS)                                                                                               \
  SPAMD_IDX_SWITCH(idx_dtype, I, hipLaunchKernelGGL((keys_to_csr_kernel<I, CLS>), dim3(grid_for(total)), dim3(256), 0, \
                                                    (hipStream_t)stream, keys, nnz, R, Cn, invC, (I*)indptr, (I*)indices))
  if (cls == 1) { SPAMD_K2C(1) } else if (cls == 2) { SPAMD_K2C(2) } else { SPAMD_K2C(0) }
#undef SPAMD_K2C
  return launch_status();
}

extern "C" int spamd_csr_to_keys(int idx_dtype, int64_t R, int64_t nnz, const void* indptr, const void* indices,
                                 int64_t C, int64_t* keys, void* stream) {
  if (nnz < 0 || R < 0) return SPAMD_EINVAL;
  if (nnz == 0) return 0;
  if (nnz >= 8 * R) {   // rows of at least eight elements on average: a wave per row
    const unsigned blocks = (unsigned)std::min<int64_t>((R + 3) / 4, (int64_t)256 * 64);
    SPAMD_IDX_SWITCH(idx_dtype, I, hipLaunchKernelGGL(csr_to_keys_rows_kernel<I>, dim3(blocks), dim3(256), 0,
                                                      (hipStream_t)stream, (const I*)indptr, (const I*)indices, R, C, keys))
    return launch_status();
  }
  SPAMD_IDX_SWITCH(idx_dtype, I, hipLaunchKernelGGL(csr_to_keys_kernel<I>, dim3(grid_for(nnz)), dim3(256), 0,
                                                    (hipStream_t)stream, (const I*)indptr, (const I*)indices, R, nnz,
                                                    C, keys))
  return launch_status();
}

extern "C" int spamd_rows_to_indptr(int idx_dtype, int64_t nnz, const void* rows, int64_t R, int64_t* indptr,
                                    void* stream) {
  if (nnz < 0 || R < 0) return SPAMD_EINVAL;
  if (nnz > 0 && R <= 8 * nnz) {   // (not hypersparse: the element-side form; empty stretches are short)
    SPAMD_IDX_SWITCH(idx_dtype, I, hipLaunchKernelGGL(rows_to_indptr_fill_kernel<I>, dim3(grid_for(nnz)), dim3(256), 0,
                                                      (hipStream_t)stream, (const I*)rows, nnz, R, indptr))
    return launch_status();
  }
  SPAMD_IDX_SWITCH(idx_dtype, I, hipLaunchKernelGGL(rows_to_indptr_kernel<I>, dim3(grid_for(R + 1)), dim3(256), 0,
                                                    (hipStream_t)stream, (const I*)rows, nnz, R, indptr))
  return launch_status();
}

// ---- rocPRIM-backed primitives (stable radix sort, exclusive scan) ---------------------------
// rocPRIM's radix sort runs a MERGE sort below `merge_sort_limit` items (default 2^20) whatever the key width.  Measured on
// MI355X for (8-byte key, 8-byte payload) pairs (tools/micro/sort_limit.hip, ms): n = 10^6: merge 0.19, radix (onesweep)
// 0.093 / 0.121 / 0.150 for 20 / 30 / 40 key bits; n = 3*10^5: 0.13 against 0.080 / 0.105 / 0.128; n = 10^5: 0.057 against
// 0.073 / 0.096 / 0.119.  BASELINE config 1 (10^6 stored elements) sits right under the default limit, so the limit is moved to
// the measured crossover: 128 K pairs for keys of at most 24 bits, 256 K above.
namespace spamd {
using SortNarrowKeys = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 128 * 1024>;
using SortWideKeys = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 256 * 1024>;

template <typename K, typename V>
static hipError_t sort_pairs_tuned(void* ws, size_t& bytes, const K* kin, K* kout, const V* vin, V* vout, size_t n, unsigned end_bit,
                                   hipStream_t s) {
  if (end_bit <= 24) return rocprim::radix_sort_pairs<SortNarrowKeys>(ws, bytes, kin, kout, vin, vout, n, 0, end_bit, s);
  return rocprim::radix_sort_pairs<SortWideKeys>(ws, bytes, kin, kout, vin, vout, n, 0, end_bit, s);
}

// workspace that serves either configuration (the key width is not known when the caller allocates)
template <typename K, typename V>
static int64_t sort_pairs_tuned_ws(int64_t n, unsigned max_bits) {
  size_t a = 0, b = 0;
  K* k = nullptr;
  V* v = nullptr;
  const size_t m = (size_t)(n > 0 ? n : 1);
  hipError_t e = sort_pairs_tuned<K, V>(nullptr, a, k, k, v, v, m, max_bits < 24 ? max_bits : 24, (hipStream_t)0);
  if (e != hipSuccess) return -(int64_t)e;
  if (max_bits > 24) {
    e = sort_pairs_tuned<K, V>(nullptr, b, k, k, v, v, m, max_bits, (hipStream_t)0);
    if (e != hipSuccess) return -(int64_t)e;
  }
  return (int64_t)(a > b ? a : b);
}
}  // namespace spamd

extern "C" int64_t spamd_sort_pairs_ws_bytes(int64_t n) {
  const int64_t bytes = spamd::sort_pairs_tuned_ws<int64_t, int64_t>(n, 64);
  return bytes < 0 ? bytes : bytes + 16;
}

// Stable LSD radix sort of (key, value) pairs on bits [0, end_bit) of the int64 keys (keys >= 0).
extern "C" int spamd_sort_pairs(int64_t n, const int64_t* keys_in, int64_t* keys_out, const int64_t* vals_in,
                                int64_t* vals_out, int end_bit, void* ws, int64_t ws_bytes, void* stream) {
  if (n < 0 || end_bit < 1 || end_bit > 64) return SPAMD_EINVAL;
  if (n == 0) return 0;
  size_t bytes = (size_t)ws_bytes;
  hipError_t e = spamd::sort_pairs_tuned(ws, bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, (unsigned)end_bit,
                                         (hipStream_t)stream);
  return (int)e;
}

// Same sort with the VALUE as payload (4- or 8-byte values moved bit-wise): SpGEMM sorts its
// (key, product) pairs directly instead of sorting a permutation and gathering through it.
extern "C" int64_t spamd_sort_kv_ws_bytes(int val_bytes, int64_t n) {
  const int64_t bytes = val_bytes == 8 ? spamd::sort_pairs_tuned_ws<int64_t, uint64_t>(n, 64)
                                       : spamd::sort_pairs_tuned_ws<int64_t, uint32_t>(n, 64);
  return bytes < 0 ? bytes : bytes + 16;
}

extern "C" int spamd_sort_kv(int val_bytes, int64_t n, const int64_t* keys_in, int64_t* keys_out, const void* vals_in,
                             void* vals_out, int end_bit, void* ws, int64_t ws_bytes, void* stream) {
  if (n < 0 || end_bit < 1 || end_bit > 64) return SPAMD_EINVAL;
  if (n == 0) return 0;
  size_t bytes = (size_t)ws_bytes;
  hipError_t e;
  if (val_bytes == 8)
    e = spamd::sort_pairs_tuned(ws, bytes, keys_in, keys_out, (const uint64_t*)vals_in, (uint64_t*)vals_out, (size_t)n,
                                (unsigned)end_bit, (hipStream_t)stream);
  else if (val_bytes == 4)
    e = spamd::sort_pairs_tuned(ws, bytes, keys_in, keys_out, (const uint32_t*)vals_in, (uint32_t*)vals_out, (size_t)n,
                                (unsigned)end_bit, (hipStream_t)stream);
  else
    return SPAMD_ETYPE;
  return (int)e;
}

extern "C" int spamd_iota(int64_t n, int64_t* out, void* stream) {
  if (n < 0) return SPAMD_EINVAL;
  if (n == 0) return 0;
  hipLaunchKernelGGL(iota_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, out, n);
  return launch_status();
}

extern "C" int64_t spamd_scan_ws_bytes(int64_t n) {
  size_t bytes = 0;
  int64_t* p = nullptr;
  hipError_t e = rocprim::exclusive_scan(nullptr, bytes, p, p, (int64_t)0, (size_t)(n > 0 ? n : 1),
                                         rocprim::plus<int64_t>(), (hipStream_t)0);
  if (e != hipSuccess) return -(int64_t)e;
  return (int64_t)bytes + 16;
}

// out[i] = sum(in[0..i)) for i in [0, n]: `out` has n+1 entries, `in` must have n+1 readable
// entries (the last one is ignored), so out[n] is the total.
extern "C" int spamd_exclusive_scan(int64_t n, const int64_t* in, int64_t* out, void* ws, int64_t ws_bytes,
                                    void* stream) {
  if (n < 0) return SPAMD_EINVAL;
  if (n + 1 <= SMALL_SCAN_MAX) {   // short arrays: one workgroup (the device-wide primitive costs ~30 us at any length)
    hipLaunchKernelGGL(small_exclusive_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, in, out, (int)(n + 1));
    return launch_status();
  }
  size_t bytes = (size_t)ws_bytes;
  hipError_t e = rocprim::exclusive_scan(ws, bytes, in, out, (int64_t)0, (size_t)(n + 1), rocprim::plus<int64_t>(),
                                         (hipStream_t)stream);
  return (int)e;
}

// dtype conversion of a value / index array (astype; C-cast semantics as NumPy's "unsafe")
extern "C" int spamd_convert(int src_dtype, int dst_dtype, int64_t n, const void* src, void* dst, void* stream) {
  if (n < 0) return SPAMD_EINVAL;
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
#define CV(SC, ST, DC, DT)                                                                                   \
  if (src_dtype == SC && dst_dtype == DC) {                                                                  \
    hipLaunchKernelGGL((convert_kernel<ST, DT>), dim3(grid_for(n)), dim3(256), 0, s, (const ST*)src, n, (DT*)dst); \
    return launch_status();                                                                                  \
  }
#define CV_ROW(SC, ST)            \
  CV(SC, ST, SPAMD_F32, float)    \
  CV(SC, ST, SPAMD_F64, double)   \
  CV(SC, ST, SPAMD_I32, int32_t)  \
  CV(SC, ST, SPAMD_I64, int64_t)  \
  CV(SC, ST, SPAMD_U8, uint8_t)
  CV_ROW(SPAMD_F32, float)
  CV_ROW(SPAMD_F64, double)
  CV_ROW(SPAMD_I32, int32_t)
  CV_ROW(SPAMD_I64, int64_t)
  CV_ROW(SPAMD_U8, uint8_t)
#undef CV_ROW
#undef CV
  return SPAMD_ETYPE;
}

// ---- A6: re-compress a 2-D compressed matrix along its other axis (CSR <-> CSC), 4- and 8-byte values ---------------
// `GCXS.change_compressed_axes` / `_transpose` (reference _compressed/compressed.py:388-423, convert.py:210-273:
// uncompress, re-linearise, stable argsort, bincount + cumsum).  The input is ordered by (major, minor), so a STABLE sort
// on the minor index alone gives (minor, major) order: ceil(log2(n_minor)) bits of a 32-bit key, and the major id rides
// along with the value bits (everything streams, nothing is gathered by a permutation).
// Rounds 2-4 packed (value, major id) into one 8- or 16-byte payload before the sort and unpacked it afterwards: two more
// passes over everything.  Late round 4: the sort reads the minor indices as its keys through a casting iterator and moves
// (value, major id) through zip iterators over the caller's own arrays - its last pass writes out_data / out_indices
// directly; only the major ids (implied by the pointers) are materialised beforehand, and the new pointers derived from
// the sorted keys afterwards.  Config 2's matrix (10^8 stored elements), CSR -> CSC (14 key bits, two passes): f32 / int32
// 2.47 -> 1.97 ms, f64 / int64 3.47 -> 2.62 ms; CSC -> CSR (20 bits, three passes, the zipped arrays are an intermediate
// buffer too): 3.11 -> 2.91 and 4.53 -> 4.30 ms (tools/r04/csx_time.py).
namespace spamd {
template <typename I>
__global__ void __launch_bounds__(256) csx_major_kernel(int64_t n_major, const I* __restrict__ indptr, I* __restrict__ major) {
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); row < n_major; row += nwaves) {
    const int64_t a = (int64_t)indptr[row], b = (int64_t)indptr[row + 1];
    for (int64_t e = a + lane; e < b; e += 64) major[e] = (I)row;
  }
}

template <typename I>
__global__ void __launch_bounds__(256) csx_pointers_kernel(int64_t nnz, int64_t n_minor, const uint32_t* __restrict__ keys,
                                                           I* __restrict__ out_indptr) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += stride) {
    // element e opens every minor index in (key[e-1], key[e]]; the last element closes the rest
    const int64_t k = (int64_t)keys[e];
    const int64_t kp = e > 0 ? (int64_t)keys[e - 1] : -1;
    for (int64_t j = kp + 1; j <= k; ++j) out_indptr[j] = (I)e;
    if (e == nnz - 1)
      for (int64_t j = k + 1; j <= n_minor; ++j) out_indptr[j] = (I)nnz;
  }
}

template <typename I>
struct CsxKeyOf {
  __host__ __device__ uint32_t operator()(const I& x) const { return (uint32_t)x; }
};

template <typename I, typename V>
static hipError_t csx_zip_sort(void* ws, size_t& bytes, const I* indices, uint32_t* keys_out, const V* data, const I* major,
                               V* out_data, I* out_indices, size_t n, unsigned bits, hipStream_t s) {
  auto kin = rocprim::make_transform_iterator(indices, CsxKeyOf<I>());
  auto vin = rocprim::make_zip_iterator(rocprim::make_tuple(data, major));
  auto vout = rocprim::make_zip_iterator(rocprim::make_tuple(out_data, out_indices));
  if (bits <= 24) return rocprim::radix_sort_pairs<SortNarrowKeys>(ws, bytes, kin, keys_out, vin, vout, n, 0, bits, s);
  return rocprim::radix_sort_pairs<SortWideKeys>(ws, bytes, kin, keys_out, vin, vout, n, 0, bits, s);
}
}  // namespace spamd

static size_t csx_align(size_t x) { return (x + 255) & ~(size_t)255; }

// workspace: sorted keys + major ids (8-byte indices at most) + rocPRIM's own double buffers (either configuration)
template <typename V>
static int64_t csx_ws_bytes(int64_t nnz) {
  const size_t n = (size_t)(nnz > 0 ? nnz : 1);
  size_t a = 0, b = 0;
  if (spamd::csx_zip_sort<int64_t, V>(nullptr, a, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, n, 24, (hipStream_t)0) != hipSuccess ||
      spamd::csx_zip_sort<int64_t, V>(nullptr, b, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, n, 32, (hipStream_t)0) != hipSuccess)
    return -1;
  return (int64_t)(csx_align(4 * n) + csx_align(8 * n) + csx_align(a > b ? a : b) + 256);
}

extern "C" int64_t spamd_csx_swap_ws_bytes(int64_t nnz) {
  if (nnz < 0) return -1;
  return csx_ws_bytes<uint32_t>(nnz);
}
extern "C" int64_t spamd_csx_swap8_ws_bytes(int64_t nnz) {
  if (nnz < 0) return -1;
  return csx_ws_bytes<uint64_t>(nnz);
}

template <typename V>
static int csx_swap(int idx_dtype, int64_t n_major, int64_t n_minor, int64_t nnz, const void* data, const void* indices,
                    const void* indptr, void* out_data, void* out_indices, void* out_indptr, void* ws, int64_t ws_bytes,
                    void* stream) {
  using namespace spamd;
  if (n_major < 0 || n_minor < 0 || nnz < 0 || n_major >= ((int64_t)1 << 32) || n_minor >= ((int64_t)1 << 32)) return SPAMD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (nnz == 0) {
    const size_t isz = idx_dtype == SPAMD_I32 ? 4 : 8;
    return (int)hipMemsetAsync(out_indptr, 0, (size_t)(n_minor + 1) * isz, s);
  }
  if (ws_bytes < csx_ws_bytes<V>(nnz)) return SPAMD_EWS;
  int bits = 1;
  while (bits < 32 && ((int64_t)1 << bits) < n_minor) ++bits;
  char* q = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
  uint32_t* ks = reinterpret_cast<uint32_t*>(q); q += csx_align(4 * (size_t)nnz);
  void* major = q; q += csx_align(8 * (size_t)nnz);
  size_t zbytes = (size_t)(reinterpret_cast<char*>(ws) + ws_bytes - q);
  const unsigned mb = (unsigned)std::min<int64_t>((n_major + 3) / 4, (int64_t)256 * 64);
  SPAMD_IDX_SWITCH(idx_dtype, I, {
    hipLaunchKernelGGL((csx_major_kernel<I>), dim3(mb ? mb : 1), dim3(256), 0, s, n_major, (const I*)indptr, (I*)major);
    if (int rc = launch_status()) return rc;
    hipError_t e = csx_zip_sort<I, V>(q, zbytes, (const I*)indices, ks, (const V*)data, (const I*)major, (V*)out_data,
                                      (I*)out_indices, (size_t)nnz, (unsigned)bits, s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((csx_pointers_kernel<I>), dim3(grid_for(nnz)), dim3(256), 0, s, nnz, n_minor, ks, (I*)out_indptr);
    return launch_status();
  })
  return SPAMD_ETYPE;
}

// data: nnz 4-byte values (moved bit-wise); indices/indptr of idx_dtype, n_major + 1 pointers; outputs: nnz values, nnz
// indices (the major ids), n_minor + 1 pointers, same dtypes.  n_major, n_minor < 2^32.
extern "C" int spamd_csx_swap(int idx_dtype, int64_t n_major, int64_t n_minor, int64_t nnz, const void* data,
                              const void* indices, const void* indptr, void* out_data, void* out_indices, void* out_indptr,
                              void* ws, int64_t ws_bytes, void* stream) {
  return csx_swap<uint32_t>(idx_dtype, n_major, n_minor, nnz, data, indices, indptr, out_data, out_indices, out_indptr, ws, ws_bytes, stream);
}
// the same for 8-byte values (float64 / int64: the reference's default value type); workspace: spamd_csx_swap8_ws_bytes
extern "C" int spamd_csx_swap8(int idx_dtype, int64_t n_major, int64_t n_minor, int64_t nnz, const void* data,
                               const void* indices, const void* indptr, void* out_data, void* out_indices, void* out_indptr,
                               void* ws, int64_t ws_bytes, void* stream) {
  return csx_swap<uint64_t>(idx_dtype, n_major, n_minor, nnz, data, indices, indptr, out_data, out_indices, out_indptr, ws, ws_bytes, stream);
}

// ---- dense -> stored elements in ONE pass (round 5) -----------------------------------------------------------------------
// `COO.from_numpy` (reference `COO.from_numpy`, sparse/numba_backend/_coo/core.py:341-384: `np.nonzero` of "not equal to the
// fill value" + a gather) was flag pass + exclusive scan + iota + two compactions: five launches and ~40 bytes of traffic per
// dense element (the flags, their offsets and the iota are 8 bytes each).  Here a tile of 2048 elements is read once, its
// elements that are not bit-identical to the fill value are counted, the tile's offset comes from a decoupled look-back over
// one state word per tile (tiles are numbered by a ticket, so a tile only waits for tiles that are running), and (index,
// value) go straight to their place: out_keys ascend.
namespace spamd {

constexpr int DN_THREADS = 256, DN_ITEMS = 8, DN_TILE = DN_THREADS * DN_ITEMS;

template <typename V>
__global__ void __launch_bounds__(DN_THREADS) dense_nonfill_kernel(int64_t n, const V* __restrict__ vals, V fill, V cmask, int64_t ntiles,
                                                                  unsigned long long* __restrict__ work,
                                                                  int64_t* __restrict__ out_keys, V* __restrict__ out_vals) {
  __shared__ long long s_blk;
  __shared__ unsigned long long s_base;
  __shared__ int wsum[DN_THREADS / 64 + 1];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) s_blk = (long long)atomicAdd(work, 1ull);
  __syncthreads();
  const int64_t blk = s_blk;
  unsigned long long* const state = work + 2;
  const int64_t i0 = blk * DN_TILE + (int64_t)tid * DN_ITEMS;
  V v[DN_ITEMS];
  unsigned keep = 0;
#pragma unroll
  for (int j = 0; j < DN_ITEMS; ++j) {
    v[j] = fill;
    if (i0 + j < n) v[j] = vals[i0 + j];
    keep |= ((V)((v[j] ^ fill) & cmask) != (V)0) ? 1u << j : 0u;   // (bit patterns; cmask without the sign bit: +-0 are one value)
  }
  const int mine = __popc(keep);
  const int incl = (int)wave_incl_scan_u32((unsigned)mine);
  if (lane == 63) wsum[wid] = incl;
  __syncthreads();
  int before = incl - mine, total = 0;
#pragma unroll
  for (int w = 0; w < DN_THREADS / 64; ++w) {
    before += w < wid ? wsum[w] : 0;
    total += wsum[w];
  }
  if (wid == 0) {
    const unsigned long long excl = lookback_exclusive(state, blk, (unsigned long long)total, lane);
    if (lane == 0) {
      s_base = excl;
      if (blk == ntiles - 1) work[1] = excl + (unsigned long long)total;   // (the last tile: the number of stored elements)
    }
  }
  __syncthreads();
  int64_t at = (int64_t)s_base + before;
#pragma unroll
  for (int j = 0; j < DN_ITEMS; ++j) {
    if ((keep >> j) & 1u) {
      out_keys[at] = i0 + j;
      out_vals[at] = v[j];
      ++at;
    }
  }
}

}  // namespace spamd

// work words of spamd_dense_nonfill for n elements
extern "C" int64_t spamd_dense_nonfill_work_words(int64_t n) {
  return n < 0 ? -1 : spamd::ceil_div(n > 0 ? n : 1, (int64_t)spamd::DN_TILE) + 2;
}

// vals[n] (val_bytes 1 / 2 / 4 / 8) -> the elements whose bits differ from `fill_bits` under `cmp_mask` (all ones: bit-identity;
// without a floating type's sign bit and fill_bits = 0: the numeric test `value != 0`): out_keys (their indices, ascending) and
// out_vals, both with room for n entries; work: spamd_dense_nonfill_work_words(n) int64 words (zeroed here), afterwards
// work[1] = the number of elements written.
extern "C" int spamd_dense_nonfill(int val_bytes, int64_t n, const void* vals, uint64_t fill_bits, uint64_t cmp_mask, int64_t* work,
                                   int64_t* out_keys, void* out_vals, void* stream) {
  using namespace spamd;
  if (n < 0 || !work) return SPAMD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int64_t ntiles = ceil_div(n > 0 ? n : 1, (int64_t)DN_TILE);
  if (ntiles >= ((int64_t)1 << 31)) return SPAMD_EINVAL;
  if (hipError_t e = hipMemsetAsync(work, 0, (size_t)(ntiles + 2) * sizeof(int64_t), s); e != hipSuccess) return (int)e;
  if (n == 0) return 0;
  unsigned long long* const w = reinterpret_cast<unsigned long long*>(work);
#define DN_GO(V)                                                                                                              \
  hipLaunchKernelGGL(dense_nonfill_kernel<V>, dim3((unsigned)ntiles), dim3(DN_THREADS), 0, s, n, (const V*)vals, (V)fill_bits, \
                     (V)cmp_mask, ntiles, w, out_keys, (V*)out_vals);                                                                      \
  return launch_status();
  switch (val_bytes) {
    case 1: { DN_GO(uint8_t) }
    case 2: { DN_GO(uint16_t) }
    case 4: { DN_GO(uint32_t) }
    case 8: { DN_GO(uint64_t) }
    default: return SPAMD_ETYPE;
  }
#undef DN_GO
}
