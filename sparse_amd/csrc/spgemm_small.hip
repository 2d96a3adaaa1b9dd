// A4 / A5 for SMALL operands (round 5): C = A @ B, both compressed by rows, in ONE launch and one read-back.
// Reference: `_csr_csr_count_nnz` + `_dot_csr_csr`, sparse/numba_backend/_common.py:543-570,639-717 - Gustavson's row loop
// with a dense accumulator `sums[n_col]` and a linked list of the touched columns.
//
// The reference's own benchmark (benchmarks/test_benchmark_coo.py:9-40: m, n, p in {200, 500, 1000}, density 0.01) multiplies
// operands of 400-10^4 stored elements.  The general path of this backend (row products, scan, bitmap / bucket kernels,
// classification, pack, prune count: 8-12 C-ABI calls and two read-backs) took 230-300 us for them - launches and host waits,
// not work; the single-core reference loop takes 24-450 us (bench_small.py).  Here the reference's loop is run as it stands,
// a WAVE per output row:
//   * the dense accumulator of the row (n_col values) and a bitmap of its touched columns live in LDS;
//   * the row's A elements are walked one after the other (the order of `sums[j] += ...`), the lanes spread over the B row of
//     each: its columns are distinct, so a step has no two lanes on one accumulator, and steps follow each other in LDS order -
//     every output element is summed left to right in A's order, bit-identical to the bucket / bitmap kernels;
//   * the row's length is the bitmap's popcount; a decoupled look-back over one state word per row (rows are numbered in
//     dispatch order) gives its offset, and the row leaves column-sorted (the bitmap's order) to its final place.
// The caller allocates the result for its upper bound (n_row x n_col entries, bounded by the size limit below) and reads
// {failed, exact zeros written, total} back once.  A B row whose columns do not ascend strictly (two products of one step could
// then hit one accumulator) or lie outside [0, n_col) sets `failed`: the caller takes the general path.
#include "common.h"

namespace spamd {

template <typename V>
__device__ __forceinline__ int sm_is_zero_bits(V v) {
  if constexpr (sizeof(V) == 8) return __builtin_bit_cast(unsigned long long, v) == 0;
  else return __builtin_bit_cast(unsigned, v) == 0;
}

// products and sums of integer values wrap around (NumPy's arithmetic): formed on the unsigned type
template <typename V>
__device__ __forceinline__ V sm_mul(V a, V b) {
  if constexpr (std::is_integral<V>::value) {
    using U = typename std::make_unsigned<V>::type;
    return (V)((U)a * (U)b);
  } else {
    return a * b;
  }
}
template <typename V>
__device__ __forceinline__ V sm_add(V a, V b) {
  if constexpr (std::is_integral<V>::value) {
    using U = typename std::make_unsigned<V>::type;
    return (V)((U)a + (U)b);
  } else {
    return a + b;
  }
}

template <typename V, typename I, int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
spgemm_small_kernel(int64_t n_row, int n_col, int words, const I* __restrict__ a_ptr, const I* __restrict__ a_idx,
                    const V* __restrict__ a_val, const I* __restrict__ b_ptr, const I* __restrict__ b_idx,
                    const V* __restrict__ b_val, unsigned long long* __restrict__ work, int64_t* __restrict__ out_ptr,
                    int64_t* __restrict__ out_idx, V* __restrict__ out_val) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) char sm_lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // Rows are numbered by a TICKET the workgroup takes when it starts (work[0], zeroed by the caller), not by blockIdx: the
  // look-back below waits for the rows before this one, which must then be running or done - HIP does not promise that
  // workgroups are dispatched in blockIdx order (round-5 advice; dense_nonfill_kernel and the merge kernels do the same).
  __shared__ unsigned long long sm_ticket;
  if (threadIdx.x == 0) sm_ticket = atomicAdd(work, 1ull);
  __syncthreads();
  const int64_t r = (int64_t)sm_ticket * WAVES + wv;
  if (r >= n_row) return;       // (wave-uniform; no workgroup barrier below)
  const size_t per_wave = ((size_t)n_col * sizeof(V) + (size_t)words * 4 + 15) / 16 * 16;
  V* const acc = reinterpret_cast<V*>(sm_lds + (size_t)wv * per_wave);
  unsigned* const bm = reinterpret_cast<unsigned*>(sm_lds + (size_t)wv * per_wave + (size_t)n_col * sizeof(V));
  unsigned long long* const state = work + 4;
  for (int w = lane; w < words; w += 64) bm[w] = 0;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  bool bad = false;
  const int64_t a0 = (int64_t)a_ptr[r], a1 = (int64_t)a_ptr[r + 1];
  for (int64_t ec = a0; ec < a1; ec += 64) {
   // 64 of the row's A elements at a time: lane l fetches element ec + l and the ends of its B row (two dependent latencies
   // per chunk instead of three per element), the elements are then taken one after the other through lane broadcasts
   const int64_t me = ec + lane;
   const bool have = me < a1;
   const int64_t mk = have ? (int64_t)a_idx[me] : 0;
   const V mav = have ? a_val[me] : V(0);
   const int64_t mb0 = have ? (int64_t)b_ptr[mk] : 0, mb1 = have ? (int64_t)b_ptr[mk + 1] : 0;
   const int nchunk = (int)(a1 - ec < 64 ? a1 - ec : 64);
   for (int el = 0; el < nchunk; ++el) {       // A's elements in storage order: the order every output element is summed in
    const V av = wave_bcast(mav, el);
    const int64_t b0 = wave_bcast(mb0, el), b1 = wave_bcast(mb1, el);
    for (int64_t j0 = b0; j0 < b1; j0 += 64) {
      const int64_t j = j0 + lane;
      if (j < b1) {
        const int64_t c = (int64_t)b_idx[j];
        const int64_t cp = j > b0 ? (int64_t)b_idx[j - 1] : -1;
        if (c <= cp || c >= n_col) {
          bad = true;
        } else {
          const V p = sm_mul(av, b_val[j]);
          const unsigned bit = 1u << (c & 31);
          const unsigned old = atomicOr(&bm[c >> 5], bit);
          acc[c] = (old & bit) ? sm_add(acc[c], p) : p;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
   }
  }
  // the row's columns in ascending order: lane l owns the bitmap words [l * wpl, (l + 1) * wpl)
  const int wpl = (words + 63) / 64;
  int mine = 0;
  for (int q = 0; q < wpl; ++q) {
    const int w = lane * wpl + q;
    mine += w < words ? __popc(bm[w]) : 0;
  }
  int incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int y = __shfl_up(incl, d, 64);
    if (lane >= d) incl += y;
  }
  const int cnt = __shfl(incl, 63, 64);
  const unsigned long long before = lookback_exclusive(state, r, (unsigned long long)cnt, lane);
  if (lane == 0) {
    out_ptr[r + 1] = (int64_t)before + cnt;
    if (r == 0) out_ptr[0] = 0;
    if (r == n_row - 1) work[3] = before + (unsigned long long)cnt;   // (the total, next to the other two words the host reads)
  }
  int64_t pos = (int64_t)before + (incl - mine);
  int zeros = 0;
  for (int q = 0; q < wpl; ++q) {
    const int w = lane * wpl + q;
    unsigned m = w < words ? bm[w] : 0u;
    while (m) {
      const int b = __builtin_ctz(m);
      m &= m - 1;
      const int c = w * 32 + b;
      const V v = acc[c];
      out_idx[pos] = c;
      out_val[pos] = v;
      zeros += sm_is_zero_bits(v);
      ++pos;
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) zeros += __shfl_xor(zeros, d, 64);
  if (lane == 0 && zeros) atomicAdd(work + 2, (unsigned long long)zeros);
  if (__ballot(bad) != 0 && lane == 0) __hip_atomic_store(work + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename V>
static int64_t sm_max_cols(int waves) {   // columns whose accumulator + bitmap fit a wave's share of 64 KB
  const int64_t per_wave = (64 * 1024) / waves - 32;
  return per_wave * 8 / (8 * (int64_t)sizeof(V) + 1);
}

template <typename V, typename I>
static int sm_launch(int64_t n_row, int64_t n_col, const I* a_ptr, const I* a_idx, const V* a_val, const I* b_ptr, const I* b_idx,
                     const V* b_val, unsigned long long* work, int64_t* out_ptr, int64_t* out_idx, V* out_val, hipStream_t s) {
  const int words = (int)ceil_div(n_col, (int64_t)32);
  const size_t per_wave = ((size_t)n_col * sizeof(V) + (size_t)words * 4 + 15) / 16 * 16;
  if (n_col <= sm_max_cols<V>(4)) {
    hipLaunchKernelGGL((spgemm_small_kernel<V, I, 4>), dim3((unsigned)ceil_div(n_row, (int64_t)4)), dim3(256), 4 * per_wave, s, n_row,
                       (int)n_col, words, a_ptr, a_idx, a_val, b_ptr, b_idx, b_val, work, out_ptr, out_idx, out_val);
  } else if (n_col <= sm_max_cols<V>(1)) {
    hipLaunchKernelGGL((spgemm_small_kernel<V, I, 1>), dim3((unsigned)n_row), dim3(64), per_wave, s, n_row, (int)n_col, words, a_ptr,
                       a_idx, a_val, b_ptr, b_idx, b_val, work, out_ptr, out_idx, out_val);
  } else {
    return SPAMD_EINVAL;
  }
  return launch_status();
}

}  // namespace spamd

using namespace spamd;

// the widest result (columns) spamd_spgemm_small takes for this value type
extern "C" int64_t spamd_spgemm_small_max_cols(int val_dtype) {
  return (val_dtype == SPAMD_F32 || val_dtype == SPAMD_I32) ? sm_max_cols<float>(1) : sm_max_cols<double>(1);
}

// C = A @ B for small operands, rows column-sorted and written in place: out_indptr[n_row + 1], out_indices / out_data with
// room for n_row * n_col entries (the caller trims to out_indptr[n_row]); work = n_row + 4 words, zeroed here; afterwards
// work[1] != 0 = failed (a B row that is not strictly ascending / in range: discard the result), work[2] = values written
// whose bits are all zero, work[3] = stored elements of the result.  n_col <= spamd_spgemm_small_max_cols(val_dtype), n_row < 2^31.
extern "C" int spamd_spgemm_small(int val_dtype, int idx_dtype, int64_t n_row, int64_t n_col, const void* a_indptr,
                                  const void* a_indices, const void* a_data, const void* b_indptr, const void* b_indices,
                                  const void* b_data, int64_t* work, int64_t* out_indptr, int64_t* out_indices, void* out_data,
                                  void* stream) {
  if (n_row < 0 || n_row >= ((int64_t)1 << 31) || n_col <= 0 || !work || !out_indptr) return SPAMD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (hipError_t e = hipMemsetAsync(work, 0, (size_t)(n_row + 4) * sizeof(int64_t), s); e != hipSuccess) return (int)e;
  if (n_row == 0) return (int)hipMemsetAsync(out_indptr, 0, sizeof(int64_t), s);
  unsigned long long* const w = reinterpret_cast<unsigned long long*>(work);
  SPAMD_DISPATCH_VAL(val_dtype, V, {
    SPAMD_DISPATCH_IDX(idx_dtype, I, {
      return (sm_launch<V, I>(n_row, n_col, (const I*)a_indptr, (const I*)a_indices, (const V*)a_data, (const I*)b_indptr,
                              (const I*)b_indices, (const V*)b_data, w, out_indptr, out_indices, (V*)out_data, s));
    })
  })
  return SPAMD_ETYPE;
}
