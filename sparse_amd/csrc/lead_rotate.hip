// A8 / A6: the keys of a canonical COO with its LEADING axes moved last, in sorted order, WITHOUT a sort (round 5).
// Reference: the reduction over axes 0 .. m-1 transposes the kept axes to the front and sorts the coordinates
// (`COO.reduce` -> `_reduce_calc`, sparse/numba_backend/_coo/core.py:693-723: `self.transpose(neg_axis + axis)`, then
// `reshape` - a sort of every stored element by its new linear index).  This backend did the same: key permutation + a
// radix sort of (key, value) pairs - 0.15 of the 0.25 ms of `sum(axis=0)` at BASELINE config 1 (10^6 elements, the row of
// `paths` furthest from its roofline).
//
// But the input is NOT arbitrary: its keys are sorted, key = s * P + c with s = the leading axes' index ("slab", S of them)
// and c = the kept axes' index ("cell", P of them), so the elements are S sorted runs - one per slab - and the wanted order
// (by c, then s) is their S-way merge.  With few slabs (S <= 2048) that merge is done by cell RANGES:
//   rl_split_kernel + rl_bounds_kernel   bnd[s][b] = first element of slab s whose cell is >= b * C (C cells per range, a power
//                     of two): short gaps by the elements themselves, the rest by a binary search per boundary word.
//   rl_merge_kernel   a workgroup per cell range: thread s takes slab s's piece [bnd[s][b], bnd[s][b + 1]) (short: ~n C / (S P)
//                     elements), the range's elements (~1000-1500) are counted per cell in LDS, placed into per-cell segments and
//                     every segment (the elements of ONE output cell: ~1, from distinct slabs) is ordered by slab by one thread.
//                     The range's first output position is the number of elements in the ranges before it = sum over the slabs
//                     of bnd[s][b] - bnd[s][0]: no scan over workgroups, no look-back.
// Measured at config 1 (10^6 elements, S = 1000, P = 10^6): boundaries ~15 us + merge 34 us against 145 us of key
// permutation + radix sort; `sum(axis=0)` 0.23 -> 0.14 ms; S = 64 .. 2000 runs and 10^5 .. 4 x 10^6 elements: 0.12-0.31 ms
// against 0.15-0.34 ms (tools/r05/sum0_shapes.py).  On the way: boundaries slab-major (every thread of a merge
// workgroup on a cache line of its own) 18 of 50 us in the first phase alone -> range-major; 1024-thread workgroups at 72
// VGPRs fit ONE per CU -> 512 threads; ranges of ~1000 elements = 977 workgroups in two rounds -> ~2000 elements, one round;
// the boundaries written by a pass over the ELEMENTS alone (9 us on dense data, but gaps filled word by word by one thread:
// 1.1 ms for 90 empty trailing runs) -> short gaps by the elements, a binary search for the words left unset.
// Output: out_keys[i] = c * S + s ascending (exactly `spamd_permute_keys` + `spamd_sort_kv`), values moved bit-wise.
// A range with more elements than the LDS arrays hold, or a cell with more than RL_MAX_PER_CELL elements, sets `failed`
// (nothing is written for that range): the caller then takes the sort.  Keys must be sorted and duplicate-free.
#include "common.h"

namespace spamd {

constexpr int RL_THREADS = 512;          // (three workgroups per CU: 1024 threads at the kernel's 72 VGPRs leave room for ONE)
constexpr int RL_MAX_SLABS = 2048;       // four per thread
constexpr int RL_SPT = RL_MAX_SLABS / RL_THREADS;
constexpr int RL_MAX_CELLS = 2048;       // cells of a range (four per thread; the LDS arrays stay below 64 KB)
constexpr int RL_CAP = 4096;             // elements of a range
constexpr int RL_MAX_PER_CELL = 64;      // elements of one output cell ordered by one thread (insertion)
constexpr int RL_SLAB_BITS = 11;

// bnd[b * S + s] = first element of run s whose cell is >= b * C (b = 0 .. nb; b = nb: the run's end), range-major so that a
// workgroup of the merge kernel reads its two rows of boundaries contiguously.  Two passes over a table preset to -1:
//   rl_split_kernel   one flat pass over the ELEMENTS: sorted keys make the flattened (run, range) index non-decreasing along
//                     them, so the words between an element's predecessor and the element itself are all "this element" -
//                     written by it when they are at most RL_GAP (dense data: 0 or 1 word per element, 9 us at config 1);
//   rl_bounds_kernel  one thread per word that is still -1 (longer gaps: empty runs, sparse runs, the table's head and tail):
//                     a binary search of the keys for s * P + min(b * C, P).
// (The element pass alone filled a gap of ANY length word by word in one thread: 1.1 ms for 90 empty trailing runs of 1000;
// the search alone costs the dense case ~45 us at 10^6 words.)
constexpr int RL_GAP = 8;

__global__ void __launch_bounds__(256) rl_split_kernel(int64_t n, const int64_t* __restrict__ keys, int64_t S, int64_t P, double rp,
                                                      int cshift, int64_t nb1, int* __restrict__ bnd) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= n) return;
  // (run, range) of a key: the run by a reciprocal multiply in double + an exact correction (keys are below 2^53)
  auto slab_of = [&](int64_t k) {
    int64_t s = (int64_t)((double)k * rp);
    int64_t r = k - s * P;
    while (r < 0) { --s; r += P; }
    while (r >= P) { ++s; r -= P; }
    return s;
  };
  const int64_t k = keys[q];
  const int64_t s = slab_of(k), b = (k - s * P) >> cshift;
  int64_t ws = 0, wb = 0;       // the first word after my predecessor's
  if (q > 0) {
    const int64_t kp = keys[q - 1];
    ws = slab_of(kp);
    wb = ((kp - ws * P) >> cshift) + 1;     // (<= nb1 - 1: a range index is below nb1 - 1)
  }
  const int64_t cnt = (s - ws) * nb1 + (b - wb) + 1;   // words from (ws, wb) to (s, b) in (run, range) order
  if (cnt <= 0 || cnt > RL_GAP) return;
  for (int64_t i = 0; i < cnt; ++i) {
    if (wb == nb1) {
      ++ws;
      wb = 0;
    }
    bnd[wb * S + ws] = (int)q;
    ++wb;
  }
}

__global__ void __launch_bounds__(256) rl_bounds_kernel(int64_t n, const int64_t* __restrict__ keys, int64_t S, int64_t P, int64_t C,
                                                       int64_t nb1, int* __restrict__ bnd) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= S * nb1 || bnd[t] != -1) return;
  const int64_t b = t / S, s = t - b * S;
  const int64_t c = b * C < P ? b * C : P;
  const int64_t target = s * P + c;
  int64_t lo = 0, hi = n;      // first index with keys[idx] >= target
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (keys[mid] < target) lo = mid + 1;
    else hi = mid;
  }
  bnd[t] = (int)lo;
}

// exclusive scan of one int per thread over the workgroup; total returned to every thread.  Two barriers.
__device__ __forceinline__ int rl_block_scan(int v, int& total, int* wsum) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int x = (int)wave_incl_scan_u32((unsigned)v);
  if (lane == 63) wsum[wid] = x;
  __syncthreads();
  if (wid == 0) {
    const int w = lane < RL_THREADS / 64 ? wsum[lane] : 0;
    const int xs = (int)wave_incl_scan_u32((unsigned)w);
    if (lane < RL_THREADS / 64) wsum[lane] = xs - w;
    if (lane == RL_THREADS / 64 - 1) wsum[RL_THREADS / 64] = xs;
  }
  __syncthreads();
  total = wsum[RL_THREADS / 64];
  const int r = x - v + wsum[wid];
  __syncthreads();   // (wsum is reused by the next scan)
  return r;
}

template <typename V>
__global__ void __launch_bounds__(RL_THREADS) rl_merge_kernel(int64_t n, const int64_t* __restrict__ keys, const V* __restrict__ vals,
                                                              int64_t S, int64_t P, int C, int64_t nb1,
                                                              const int* __restrict__ bnd, int64_t* __restrict__ out_keys,
                                                              V* __restrict__ out_vals, int64_t* __restrict__ failed) {
  __shared__ int cell_off[RL_MAX_CELLS + 1];
  __shared__ int cursor[RL_MAX_CELLS];
  __shared__ unsigned seg_ck[RL_CAP];    // (cell << RL_SLAB_BITS) | slab
  __shared__ unsigned seg_src[RL_CAP];   // the element's index in the input
  __shared__ int wsum[RL_THREADS / 64 + 1];
  __shared__ long long wbase[RL_THREADS / 64];
  __shared__ int bad;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t b = blockIdx.x;
  const int64_t c0 = b * C;
  // ---- the pieces of my slabs (RL_SPT consecutive ones, so that a scan over the threads follows the slab order) -----------
  int lo[RL_SPT], len[RL_SPT];
  long long before = 0;
  int mine = 0;
#pragma unroll
  for (int u = 0; u < RL_SPT; ++u) {
    const int64_t s = (int64_t)tid * RL_SPT + u;
    lo[u] = 0;
    len[u] = 0;
    if (s < S) {
      lo[u] = bnd[b * S + s];
      len[u] = bnd[(b + 1) * S + s] - lo[u];
      before += lo[u] - bnd[s];
      mine += len[u];
    }
  }
  // the first two keys of every piece, requested together (a piece holds ~1 element; one load per element inside the two loops
  // below is a chain of dependent memory latencies per thread)
  int64_t kq[RL_SPT][2];
#pragma unroll
  for (int u = 0; u < RL_SPT; ++u) {
    kq[u][0] = len[u] > 0 ? keys[lo[u]] : 0;
    kq[u][1] = len[u] > 1 ? keys[lo[u] + 1] : 0;
  }
  if (tid == 0) bad = 0;
  for (int i = tid; i <= RL_MAX_CELLS; i += RL_THREADS) cell_off[i] = 0;
  int T;
  (void)rl_block_scan(mine, T, wsum);   // (also the barrier behind the zeroing)
  // elements of the ranges before this one
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) before += __shfl_xor(before, d, 64);
  if (lane == 0) wbase[wid] = before;
  __syncthreads();
  long long out_base = 0;
#pragma unroll
  for (int w = 0; w < RL_THREADS / 64; ++w) out_base += wbase[w];
  if (T > RL_CAP) {   // (workgroup-uniform)
    if (tid == 0) atomicExch(reinterpret_cast<unsigned long long*>(failed), 1ull);
    return;
  }
  // ---- count per cell, offsets, placement into the cells' segments ----------------------------------------------------------
#pragma unroll
  for (int u = 0; u < RL_SPT; ++u) {
    const int64_t s = (int64_t)tid * RL_SPT + u;
    for (int q = 0; q < len[u]; ++q) {
      const int64_t k = q == 0 ? kq[u][0] : (q == 1 ? kq[u][1] : keys[lo[u] + q]);
      atomicAdd(&cell_off[(int)(k - s * P - c0)], 1);
    }
  }
  __syncthreads();
  {
    constexpr int CPT = RL_MAX_CELLS / RL_THREADS;
    int c4[CPT], sum = 0, mx = 0;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
      c4[k] = cell_off[tid * CPT + k];
      sum += c4[k];
      mx = c4[k] > mx ? c4[k] : mx;
    }
    if (mx > RL_MAX_PER_CELL) bad = 1;
    int tot;
    int run = rl_block_scan(sum, tot, wsum);
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
      cell_off[tid * CPT + k] = run;
      cursor[tid * CPT + k] = run;
      run += c4[k];
    }
    if (tid == RL_THREADS - 1) cell_off[RL_MAX_CELLS] = run;
  }
  __syncthreads();
  if (bad) {   // (workgroup-uniform: read behind the barrier)
    if (tid == 0) atomicExch(reinterpret_cast<unsigned long long*>(failed), 1ull);
    return;
  }
#pragma unroll
  for (int u = 0; u < RL_SPT; ++u) {
    const int64_t s = (int64_t)tid * RL_SPT + u;
    for (int q = 0; q < len[u]; ++q) {
      const int64_t k = q == 0 ? kq[u][0] : (q == 1 ? kq[u][1] : keys[lo[u] + q]);
      const int cell = (int)(k - s * P - c0);
      const int at = atomicAdd(&cursor[cell], 1);
      seg_ck[at] = ((unsigned)cell << RL_SLAB_BITS) | (unsigned)s;
      seg_src[at] = (unsigned)(lo[u] + q);
    }
  }
  __syncthreads();
  // ---- every cell's elements in slab order (they arrived in any order): one thread per cell, insertion ---------------------
#pragma unroll
  for (int k = 0; k < RL_MAX_CELLS / RL_THREADS; ++k) {
    const int c = k * RL_THREADS + tid;     // (interleaved: neighbouring lanes read neighbouring words)
    const int a = cell_off[c], e = cell_off[c + 1];
    for (int i = a + 1; i < e; ++i) {
      const unsigned ck = seg_ck[i], src = seg_src[i];
      int j = i - 1;
      while (j >= a && seg_ck[j] > ck) {
        seg_ck[j + 1] = seg_ck[j];
        seg_src[j + 1] = seg_src[j];
        --j;
      }
      seg_ck[j + 1] = ck;
      seg_src[j + 1] = src;
    }
  }
  __syncthreads();
  for (int p0 = tid; p0 < T; p0 += 4 * RL_THREADS) {   // (four gathers in flight per thread)
    V v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int p = p0 + r * RL_THREADS;
      v[r] = p < T ? vals[seg_src[p]] : V(0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int p = p0 + r * RL_THREADS;
      if (p < T) {
        const unsigned ck = seg_ck[p];
        out_keys[out_base + p] = (c0 + (int64_t)(ck >> RL_SLAB_BITS)) * S + (int64_t)(ck & ((1u << RL_SLAB_BITS) - 1u));
        out_vals[out_base + p] = v[r];
      }
    }
  }
}

}  // namespace spamd

using namespace spamd;

// limits of spamd_keys_lead_last: which = 0 slabs (S), 1 cells per range (a power of two at most this), 2 elements per range
extern "C" int64_t spamd_keys_lead_last_limits(int which) {
  switch (which) {
    case 0: return RL_MAX_SLABS;
    case 1: return RL_MAX_CELLS;
    case 2: return RL_CAP;
    default: return -1;
  }
}

// keys[n] sorted, duplicate-free, key = s * P + c (0 <= s < S, 0 <= c < P); out_keys[n] = c * S + s ascending, out_vals the
// values in that order (val_bytes 4 or 8, moved bit-wise).  cells_per_range: a power of two <= limit 1; bounds: workspace of
// S * (ceil(P / cells_per_range) + 1) ints.  *failed (device int64, zeroed here) != 0 afterwards: a range or a cell was too full,
// out_* are incomplete - use the sort.  n < 2^31, S <= limit 0.
extern "C" int spamd_keys_lead_last(int val_bytes, int64_t n, const int64_t* keys, const void* vals, int64_t S, int64_t P,
                                    int64_t cells_per_range, int* bounds, int64_t* out_keys, void* out_vals, int64_t* failed,
                                    void* stream) {
  if (n < 0 || n >= ((int64_t)1 << 31) || S < 1 || S > RL_MAX_SLABS || P < 1 || P >= ((int64_t)1 << 42) || !failed) return SPAMD_EINVAL;
  if (cells_per_range < 1 || cells_per_range > RL_MAX_CELLS || (cells_per_range & (cells_per_range - 1))) return SPAMD_EINVAL;
  if (val_bytes != 4 && val_bytes != 8) return SPAMD_ETYPE;
  hipStream_t s = (hipStream_t)stream;
  if (hipError_t e = hipMemsetAsync(failed, 0, sizeof(int64_t), s); e != hipSuccess) return (int)e;
  if (n == 0) return 0;
  const int64_t nb = ceil_div(P, cells_per_range), nb1 = nb + 1;
  if (nb >= ((int64_t)1 << 31)) return SPAMD_EINVAL;
  int cshift = 0;
  while (((int64_t)1 << cshift) < cells_per_range) ++cshift;
  if (hipError_t e = hipMemsetAsync(bounds, 0xff, (size_t)(S * nb1) * sizeof(int), s); e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(rl_split_kernel, dim3((unsigned)ceil_div(n, (int64_t)256)), dim3(256), 0, s, n, keys, S, P, 1.0 / (double)P, cshift, nb1,
                     bounds);
  if (int rc = launch_status()) return rc;
  hipLaunchKernelGGL(rl_bounds_kernel, dim3((unsigned)ceil_div(S * nb1, (int64_t)256)), dim3(256), 0, s, n, keys, S, P, cells_per_range, nb1,
                     bounds);
  if (int rc = launch_status()) return rc;
  if (val_bytes == 4)
    hipLaunchKernelGGL(rl_merge_kernel<uint32_t>, dim3((unsigned)nb), dim3(RL_THREADS), 0, s, n, keys, (const uint32_t*)vals, S, P,
                       (int)cells_per_range, nb1, (const int*)bounds, out_keys, (uint32_t*)out_vals, failed);
  else
    hipLaunchKernelGGL(rl_merge_kernel<uint64_t>, dim3((unsigned)nb), dim3(RL_THREADS), 0, s, n, keys, (const uint64_t*)vals, S, P,
                       (int)cells_per_range, nb1, (const int*)bounds, out_keys, (uint64_t*)out_vals, failed);
  return launch_status();
}
