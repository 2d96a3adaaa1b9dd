// A7: sorted-coordinate MERGE of two canonical (sorted, duplicate-free) key arrays, fused with the
// elementwise function and the fill-value prune (reference `_Elemwise.get_result` +
// `_match_arrays`, sparse/numba_backend/_umath.py:53-92,457-503,576-654).
//
// Merge-path formulation (one O(n) streaming pass instead of the reference's argsort + join +
// concatenate + re-sort, and instead of two binary searches per element):
//   partition : block p owns merged ranks [p*TILE, (p+1)*TILE); a binary search on that diagonal
//               gives where its A and B segments start.
//   count/fill: the block stages its two segments (keys + values, <= TILE items, plus one
//               look-behind A key and one look-ahead B key) in LDS; every thread finds its own
//               diagonal inside the tile and merges VT items sequentially.  A key present in both
//               operands is emitted once, by the thread that takes it from A (ties go A-first, so
//               its partner is the next B key); out = func(a or fill_a, b or fill_b); results that
//               are bit-identical to func(fill_a, fill_b) are dropped.  `count` returns per-block
//               totals (host: exclusive scan), `fill` repeats the merge and writes keys/values.
// Output keys are strictly increasing by construction: the result is canonical, no re-sort.
#include "common.h"

namespace spamd {

#ifndef SPAMD_MP_THREADS
#define SPAMD_MP_THREADS 1024
#endif
#ifndef SPAMD_MP_VT
#define SPAMD_MP_VT 4
#endif
#ifndef SPAMD_MP_ABL
#define SPAMD_MP_ABL 0   // timing ablations of mp_union_kernel (wrong results): 1 no stores, 2 no merge, 3 no look-back
#endif
constexpr int MP_THREADS = SPAMD_MP_THREADS;
constexpr int MP_VT = SPAMD_MP_VT;
constexpr int MP_TILE = MP_THREADS * MP_VT;

// ---- the elementwise functions (same codes as spamd_ewise_binary) --------------------------------
template <typename T>
__device__ __forceinline__ T mp_max(T a, T b) {
  if constexpr (std::is_floating_point<T>::value) return (a != a) ? a : ((b != b) ? b : (a > b ? a : b));
  else return a > b ? a : b;
}
template <typename T>
__device__ __forceinline__ T mp_min(T a, T b) {
  if constexpr (std::is_floating_point<T>::value) return (a != a) ? a : ((b != b) ? b : (a < b ? a : b));
  else return a < b ? a : b;
}

template <typename T, typename O>
__device__ __forceinline__ O mp_apply(int op, T a, T b) {
#pragma clang fp contract(off)
  if constexpr (std::is_same<O, uint8_t>::value) {
    // comparisons / logical ops produce 0/1 bytes whatever the operand type (T may itself be uint8_t: bool operands)
    switch (op) {
      case 32: return a > b;
      case 33: return a >= b;
      case 34: return a < b;
      case 35: return a <= b;
      case 36: return a == b;
      case 37: return a != b;
      case 38: return (a != T(0)) && (b != T(0));
      case 39: return (a != T(0)) || (b != T(0));
      case 40: return (a != T(0)) != (b != T(0));
    }
  }
  if constexpr (std::is_same<O, T>::value) {
    switch (op) {
      case 0: return a + b;
      case 1: return a - b;
      case 2: return a * b;
      case 3:
        if constexpr (std::is_floating_point<T>::value) return a / b;
        else return b == 0 ? T(0) : a / b;
      case 4: return mp_max(a, b);
      case 5: return mp_min(a, b);
      case 7:
        if constexpr (std::is_floating_point<T>::value) return (a != a) ? b : ((b != b) ? a : (a > b ? a : b));
        else return a > b ? a : b;
      case 8:
        if constexpr (std::is_floating_point<T>::value) return (a != a) ? b : ((b != b) ? a : (a < b ? a : b));
        else return a < b ? a : b;
      case 64: if constexpr (std::is_integral<T>::value) return a & b; else return T(0);
      case 65: if constexpr (std::is_integral<T>::value) return a | b; else return T(0);
      case 66: if constexpr (std::is_integral<T>::value) return a ^ b; else return T(0);
    }
    return T(0);
  } else {
    return O(0);
  }
}

template <typename O>
__device__ __forceinline__ bool mp_same_bits(O x, O y) {
  if constexpr (sizeof(O) == 8) return __builtin_bit_cast(uint64_t, x) == __builtin_bit_cast(uint64_t, y);
  else if constexpr (sizeof(O) == 4) return __builtin_bit_cast(uint32_t, x) == __builtin_bit_cast(uint32_t, y);
  else return x == y;
}

__global__ void __launch_bounds__(256) mp_partition_kernel(const int64_t* __restrict__ a, int64_t na,
                                                           const int64_t* __restrict__ b, int64_t nb,
                                                           int64_t nblocks, int64_t* __restrict__ part) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p > nblocks) return;
  int64_t d = p * MP_TILE;
  if (d > na + nb) d = na + nb;
  int64_t lo = d > nb ? d - nb : 0, hi = d < na ? d : na;
  while (lo < hi) {  // largest i with a[i-1] <= b[d-i]  (ties: A first)
    const int64_t mid = (lo + hi) >> 1;
    if (a[mid] <= b[d - 1 - mid]) lo = mid + 1; else hi = mid;
  }
  part[p] = lo;
}

// The same search by the 64 lanes of one wave, 64 probes per round: 4 rounds of two dependent loads for 10^7 keys
// instead of ~23 (the fused single-launch form below runs it at the start of every tile).
__device__ __forceinline__ int64_t mp_diag_wave(const int64_t* __restrict__ a, int64_t na, const int64_t* __restrict__ b,
                                                int64_t nb, int64_t d, int lane) {
  int64_t lo = d > nb ? d - nb : 0, hi = d < na ? d : na;
  while (lo < hi) {   // invariant: the answer (number of i with a[i] <= b[d-1-i], a prefix-true predicate) lies in [lo, hi]
    const int64_t step = (hi - lo + 63) / 64;
    const int64_t p = lo + lane * step;
    bool t = false;
    if (p < hi) t = a[p] <= b[d - 1 - p];
    const int c = __popcll(__ballot(t));   // probes 0 .. c-1 hold, probe c does not (or lies past the range)
    const int64_t nlo = c > 0 ? lo + (c - 1) * step + 1 : lo;
    const int64_t nhi = lo + c * step < hi ? lo + c * step : hi;
    lo = nlo;
    hi = nhi;
  }
  return lo;
}

// MODE 0: counts[block] = number of outputs.  MODE 1: write them at offs[block] + local rank.
// MODE 2: single pass — tiles are taken in ticket order and every tile obtains its output offset by looking back
// at its predecessors' published totals (`counts` then holds nblocks state words, zero-initialised, + the ticket
// counter + the grand total): the count pass (a full extra read of both operands) disappears; the outputs must
// have room for na + nb elements.
// MODE 3: MODE 2 in ONE launch with nothing to prepare: every tile finds its own two diagonals (mp_diag_wave: no partition
// kernel), and the workspace `counts` = [ticket, done, unused, state words ...] is left ZERO by the kernel itself (the
// last tile to finish its look-back clears it: no memset before the next call).  The number of outputs goes to
// offs[0] (device) and, when given, to *out_total_host - pinned host memory the caller spins on instead of copying back.
// VT items per thread (tile = MP_THREADS * VT).  Rounds 1-3: 512 threads x 8 items.  Round 4: 1024 threads x 4 items - the
// same 4096-item tile (same LDS, two workgroups per CU) with twice the waves, so the chain ticket -> diagonals -> segment
// loads -> look-back of a tile has 8 waves per SIMD to hide behind instead of 4 and the serial part of the merge is half as
// long (float64, 10^8 + 10^8 items: x + y 2.38 -> 2.01 ms, x * y 1.89 -> 1.53 ms; config 1 0.060 -> 0.058 ms).  Measured
// and worse, same two rows: 512 x 4 (2048-item tiles, the same wave count: 2.44 / 1.85 - the per-tile chain is what costs),
// 256 x 8 2.80 / 2.07, 256 x 4 3.05 / 2.88, 1024 x 8 (one workgroup per CU) 2.72 / 1.91, 1024 x 6 2.69 / 2.03,
// 1024 x 2 2.68 / 2.14 (tools/r04/run_mp.sh with -DSPAMD_MP_THREADS / -DSPAMD_MP_VT).
// Where the rest goes (-DSPAMD_MP_ABL, `merge_union` wall clock on 10^8 + 10^8 float64 items, 1.81 ms): without the merge
// (segments copied straight through) 1.87, without the stores 1.40, without the look-back (static offsets) 1.35 - the serial
// merge is free, the launch is the two streams plus the wait for the predecessors' totals.  Built on that and measured
// worse (all bit-identical to the two-pass form, tools/r04/merge_stream_check.py; the kernels are not kept):
//  * a persistent form (two resident workgroups per CU, the next tile's segments requested into registers one tile ahead,
//    its diagonals two ahead): tickets taken ahead put a tile that is only being prefetched in front of tiles others are
//    merging (3.8 ms); tiles dealt round-robin instead: 3.0 ms at 64 VGPRs (25 spilled dwords), 1.88 ms with one 1024-thread
//    workgroup per CU at 97 VGPRs, 2.34 ms as 512 x 8 - the prefetch buys what the second workgroup per CU bought, not more;
//  * 4 / 8 / 16 look-back windows per round trip instead of one: 2.26 / 2.29 / 2.36 ms (more state words polled by 512
//    workgroups at once; a prefix is usually found in the first window anyway).
template <typename T, typename O, int MODE, int VT = MP_VT>
__global__ void __launch_bounds__(MP_THREADS)
mp_union_kernel(int op, const int64_t* __restrict__ ka, const T* __restrict__ va, int64_t na,
                const int64_t* __restrict__ kb, const T* __restrict__ vb, int64_t nb, T fill_a, T fill_b,
                O fill_out, const int64_t* __restrict__ part, int64_t* __restrict__ counts,
                int64_t* __restrict__ offs, int64_t* __restrict__ out_keys, O* __restrict__ out_vals,
                int64_t* __restrict__ out_total_host) {
  constexpr int TILE = MP_THREADS * VT;
  __shared__ int64_t sk[TILE + 4];
  __shared__ T sv[TILE + 4];
  __shared__ int wave_tot[MP_THREADS / 64];
  constexpr bool FILL = MODE != 0;
  const int tid = threadIdx.x;
  int64_t blk = blockIdx.x;
  const int64_t nblocks = gridDim.x;
  unsigned long long* const states =
      reinterpret_cast<unsigned long long*>(MODE == 3 ? counts + 3 : counts);   // look-back state words
  if constexpr (MODE == 2 || MODE == 3) {  // ticket order = start order: a tile only ever waits for tiles that are already running
    __shared__ int64_t ticket;
    if (tid == 0) ticket = (int64_t)atomicAdd(reinterpret_cast<unsigned long long*>(MODE == 3 ? counts : counts + nblocks), 1ull);
    __syncthreads();
    blk = ticket;
  }
  int64_t d0 = blk * TILE, d1 = (blk + 1) * TILE;
  if (d1 > na + nb) d1 = na + nb;
  int64_t a0, a1;
  if constexpr (MODE == 3) {
    __shared__ int64_t diag_s[2];
    if (tid < 128) {   // wave 0: this tile's first diagonal, wave 1: its last
      const int64_t r = mp_diag_wave(ka, na, kb, nb, tid < 64 ? d0 : d1, tid & 63);
      if ((tid & 63) == 0) diag_s[tid >> 6] = r;
    }
    __syncthreads();
    a0 = diag_s[0];
    a1 = diag_s[1];
  } else {
    a0 = part[blk];
    a1 = part[blk + 1];
  }
  const int64_t b0 = d0 - a0, b1 = d1 - a1;
  const int la = (int)(a1 - a0), lb = (int)(b1 - b0);
  // LDS layout: [0] = a[a0-1] | A segment [1 .. la] | B segment [la+1 .. la+lb] | [la+lb+1] = b[b1]
  for (int t = tid; t < la + lb + 2; t += MP_THREADS) {
    int64_t k;
    T v = T(0);
    if (t == 0) k = a0 > 0 ? ka[a0 - 1] : (int64_t)-1;
    else if (t <= la) { k = ka[a0 + t - 1]; v = va[a0 + t - 1]; }
    else if (t <= la + lb) { k = kb[b0 + (t - la - 1)]; v = vb[b0 + (t - la - 1)]; }
    else k = b1 < nb ? kb[b1] : INT64_MAX;
    sk[t] = k;
    sv[t] = v;
  }
  __syncthreads();
  const int64_t* A = sk + 1;        // A[i], i in [-1, la)
  const int64_t* B = sk + 1 + la;   // B[j], j in [0, lb]
  const T* AV = sv + 1;
  const T* BV = sv + 1 + la;
  // this thread's diagonal inside the tile
  int diag = tid * VT;
  const int total = la + lb;
  if (diag > total) diag = total;
  int lo = diag > lb ? diag - lb : 0, hi = diag < la ? diag : la;
  while (SPAMD_MP_ABL != 2 && lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (A[mid] <= B[diag - 1 - mid]) lo = mid + 1; else hi = mid;
  }
  int i = lo, j = diag - lo;
  int64_t okey[VT];
  O oval[VT];
  int cnt = 0;
#if SPAMD_MP_ABL == 2
  for (int s = 0; s < VT; ++s) if (tid * VT + s < total) { okey[cnt] = A[tid * VT + s]; oval[cnt] = (O)AV[tid * VT + s]; ++cnt; }
#endif
#pragma unroll
  for (int s = 0; s < VT; ++s) {
    if (SPAMD_MP_ABL != 2 && diag + s < total) {
      const bool takeA = (i < la) && (j >= lb || A[i] <= B[j]);
      if (takeA) {
        const int64_t k = A[i];
        const bool matched = (B[j] == k);  // B[lb] is the look-ahead key (or INT64_MAX)
        T bvv = fill_b;
        if (matched) bvv = (j < lb) ? BV[j] : vb[b1];
        const O r = mp_apply<T, O>(op, AV[i], bvv);
        if (!mp_same_bits(r, fill_out)) { okey[cnt] = k; oval[cnt] = r; ++cnt; }
        ++i;
      } else {
        const int64_t k = B[j];
        if (A[i - 1] != k) {  // A[-1] is the look-behind key (or -1): a matched B key was emitted with its A
          const O r = mp_apply<T, O>(op, fill_a, BV[j]);
          if (!mp_same_bits(r, fill_out)) { okey[cnt] = k; oval[cnt] = r; ++cnt; }
        }
        ++j;
      }
    }
  }
  // block-wide exclusive scan of cnt (wave shuffles + one LDS hop)
  const int lane = tid & 63, wv = tid >> 6;
  int incl = cnt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int n = __shfl_up(incl, off, 64);
    if (lane >= off) incl += n;
  }
  if (lane == 63) wave_tot[wv] = incl;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < MP_THREADS / 64; ++w) {
    if (w < wv) base += wave_tot[w];
    tot += wave_tot[w];
  }
  if constexpr (!FILL) {
    if (tid == 0) counts[blk] = tot;
  } else {
    // stage the tile's outputs in LDS (the input staging area is dead now) and copy them out with consecutive lanes on
    // consecutive elements: writing each thread's run of up to MP_VT outputs straight from registers puts every lane
    // of a store instruction on a different cache line (measured 2.4 ms instead of 1.6 ms for 1.9e8 outputs)
    // (sparse outputs — e.g. a product of two disjoint-ish operands — are not worth the two extra barriers)
    static_assert(sizeof(O) <= sizeof(T), "the output staging area reuses sv");
    O* const so = reinterpret_cast<O*>(sv);
    const int lbase = base + (incl - cnt);
    int64_t o;
    if constexpr (MODE == 1) {
      o = offs[blk];
    } else {
      __shared__ int64_t excl_s;
      if (tid < 64) {  // wave 0 looks back 64 predecessors at a time
        const unsigned long long excl = SPAMD_MP_ABL == 3 ? (unsigned long long)(blk * TILE)
                                                          : lookback_exclusive(states, blk, (unsigned long long)tot, tid);
        if (tid == 0) {
          if (blk == nblocks - 1) {
            const int64_t total = (int64_t)(excl + (unsigned long long)tot);
            if constexpr (MODE == 3) {
              offs[0] = total;
              if (out_total_host) __hip_atomic_store(out_total_host, total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            } else {
              counts[nblocks + 1] = total;
            }
          }
          excl_s = (int64_t)excl;
        }
      }
      __syncthreads();
      o = excl_s;
      if constexpr (MODE == 3) {
        // every tile reports once its look-back is over; the last one to report knows that nobody reads the state words
        // any more and leaves the workspace zeroed for the next call
        __shared__ int last_s;
        if (tid == 0) {
          __threadfence();
          last_s = atomicAdd(reinterpret_cast<unsigned long long*>(counts + 1), 1ull) == (unsigned long long)(nblocks - 1);
        }
        __syncthreads();
        if (last_s) {
          for (int64_t t = tid; t < nblocks; t += MP_THREADS) states[t] = 0;
          if (tid == 0) counts[0] = counts[1] = 0;
        }
      }
    }
    if (tot * 4 >= TILE) {
      __syncthreads();  // everyone is done reading sk / sv
#pragma unroll
      for (int s = 0; s < VT; ++s) {
        if (s < cnt) {
          sk[lbase + s] = okey[s];
          so[lbase + s] = oval[s];
        }
      }
      __syncthreads();
      for (int t = tid; t < tot; t += MP_THREADS) {
        if (SPAMD_MP_ABL == 1 && sk[t] != -12345) continue;
        out_keys[o + t] = sk[t];
        out_vals[o + t] = so[t];
      }
    } else {
#pragma unroll
      for (int s = 0; s < VT; ++s) {
        if (s < cnt) {
          out_keys[o + lbase + s] = okey[s];
          out_vals[o + lbase + s] = oval[s];
        }
      }
    }
  }
}

template <typename T>
static T from_bits(uint64_t bits) {
  if constexpr (sizeof(T) == 8) return __builtin_bit_cast(T, bits);
  else if constexpr (sizeof(T) == 4) return __builtin_bit_cast(T, (uint32_t)bits);
  else return (T)bits;
}

}  // namespace spamd

using namespace spamd;

extern "C" int64_t spamd_merge_num_blocks(int64_t na, int64_t nb) {
  const int64_t t = na + nb;
  return t <= 0 ? 0 : (t + MP_TILE - 1) / MP_TILE;
}

// tiles of the fused form (spamd_merge_union_fused)
extern "C" int64_t spamd_merge_fused_blocks(int64_t na, int64_t nb) { return spamd_merge_num_blocks(na, nb); }

extern "C" int spamd_merge_partition(int64_t na, const int64_t* ka, int64_t nb, const int64_t* kb, int64_t* part,
                                     void* stream) {
  if (na < 0 || nb < 0) return SPAMD_EINVAL;
  const int64_t nblocks = spamd_merge_num_blocks(na, nb);
  if (nblocks == 0) return 0;
  hipLaunchKernelGGL(mp_partition_kernel, dim3((unsigned)ceil_div(nblocks + 1, 256)), dim3(256), 0,
                     (hipStream_t)stream, ka, na, kb, nb, nblocks, part);
  return launch_status();
}

// fill == 0: counts[nblocks] <- outputs per block.  fill == 1: offsets[nblocks] (exclusive scan of the
// counts) -> out_keys / out_vals.  fill == 2: single pass; counts = nblocks + 2 int64 of workspace (zeroed here),
// counts[nblocks + 1] receives the number of outputs, out_keys / out_vals hold na + nb elements.
// val_dtype F32|F64|I32|I64|U8; comparisons/logical ops write U8.
extern "C" int spamd_merge_union(int fill, int op, int val_dtype, int64_t na, const int64_t* ka, const void* va,
                                 int64_t nb, const int64_t* kb, const void* vb, uint64_t fill_a_bits,
                                 uint64_t fill_b_bits, uint64_t fill_out_bits, const int64_t* part,
                                 int64_t* counts, const int64_t* offsets, int64_t* out_keys, void* out_vals,
                                 void* stream) {
  if (na < 0 || nb < 0) return SPAMD_EINVAL;
  const int64_t nblocks = spamd_merge_num_blocks(na, nb);
  if (nblocks == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (fill == 2) {
    hipError_t e = hipMemsetAsync(counts, 0, (size_t)(nblocks + 2) * sizeof(int64_t), s);
    if (e != hipSuccess) return (int)e;
  }
  const bool to_bool = op >= 32 && op < 64;
  if (op >= 64 && (val_dtype == SPAMD_F32 || val_dtype == SPAMD_F64)) return SPAMD_ETYPE;
  if (op == 6) return SPAMD_ETYPE;  // power: use the aligned-array path
#define MP_LAUNCH(T, O)                                                                                       \
  do {                                                                                                        \
    if (fill == 2)                                                                                     \
      hipLaunchKernelGGL((mp_union_kernel<T, O, 2>), dim3((unsigned)nblocks), dim3(MP_THREADS), 0, s, op, ka,    \
                         (const T*)va, na, kb, (const T*)vb, nb, from_bits<T>(fill_a_bits),                   \
                         from_bits<T>(fill_b_bits), from_bits<O>(fill_out_bits), part, counts,                \
                         const_cast<int64_t*>(offsets), out_keys, (O*)out_vals, (int64_t*)nullptr);           \
    else if (fill)                                                                                            \
      hipLaunchKernelGGL((mp_union_kernel<T, O, 1>), dim3((unsigned)nblocks), dim3(MP_THREADS), 0, s, op, ka,    \
                         (const T*)va, na, kb, (const T*)vb, nb, from_bits<T>(fill_a_bits),                   \
                         from_bits<T>(fill_b_bits), from_bits<O>(fill_out_bits), part, counts,                \
                         const_cast<int64_t*>(offsets), out_keys, (O*)out_vals, (int64_t*)nullptr);           \
    else                                                                                                      \
      hipLaunchKernelGGL((mp_union_kernel<T, O, 0>), dim3((unsigned)nblocks), dim3(MP_THREADS), 0, s, op, ka,    \
                         (const T*)va, na, kb, (const T*)vb, nb, from_bits<T>(fill_a_bits),                   \
                         from_bits<T>(fill_b_bits), from_bits<O>(fill_out_bits), part, counts,                \
                         const_cast<int64_t*>(offsets), out_keys, (O*)out_vals, (int64_t*)nullptr);           \
  } while (0)
  switch (val_dtype) {
    case SPAMD_F32: if (to_bool) MP_LAUNCH(float, uint8_t); else MP_LAUNCH(float, float); break;
    case SPAMD_F64: if (to_bool) MP_LAUNCH(double, uint8_t); else MP_LAUNCH(double, double); break;
    case SPAMD_I32: if (to_bool) MP_LAUNCH(int32_t, uint8_t); else MP_LAUNCH(int32_t, int32_t); break;
    case SPAMD_I64: if (to_bool) MP_LAUNCH(int64_t, uint8_t); else MP_LAUNCH(int64_t, int64_t); break;
    case SPAMD_U8: MP_LAUNCH(uint8_t, uint8_t); break;
    default: return SPAMD_ETYPE;
  }
#undef MP_LAUNCH
  return launch_status();
}

// The fused single-launch form (mp_union_kernel MODE 3): no partition kernel, no memset, no copy-back.
//   ws         : int64[3 + capacity] device workspace, capacity >= spamd_merge_fused_blocks(na, nb); all ZERO before the
//                first use; every call leaves it zero again (calls sharing a workspace must be stream-ordered)
//   total_dev  : device int64 that receives the number of outputs
//   total_host : null, or pinned host memory mapped to the device (hipHostMalloc / torch pin_memory): receives the same
//                number with a system-scope release store as soon as it is known, so the host may spin on it instead
//                of synchronising the stream
//   out_keys / out_vals hold na + nb elements.
extern "C" int spamd_merge_union_fused(int op, int val_dtype, int64_t na, const int64_t* ka, const void* va, int64_t nb,
                                       const int64_t* kb, const void* vb, uint64_t fill_a_bits, uint64_t fill_b_bits,
                                       uint64_t fill_out_bits, int64_t* ws, int64_t* total_dev, int64_t* total_host,
                                       int64_t* out_keys, void* out_vals, void* stream) {
  if (na < 0 || nb < 0 || !ws || !total_dev) return SPAMD_EINVAL;
  const int64_t nblocks = spamd_merge_fused_blocks(na, nb);
  if (nblocks == 0) return SPAMD_EINVAL;   // (nothing to merge: the caller returns an empty result without a launch)
  hipStream_t s = (hipStream_t)stream;
  const bool to_bool = op >= 32 && op < 64;
  if (op >= 64 && (val_dtype == SPAMD_F32 || val_dtype == SPAMD_F64)) return SPAMD_ETYPE;
  if (op == 6) return SPAMD_ETYPE;  // power: use the aligned-array path
#define MP_FUSED(T, O)                                                                                          \
  hipLaunchKernelGGL((mp_union_kernel<T, O, 3>), dim3((unsigned)nblocks), dim3(MP_THREADS), 0, s, op, ka,        \
                     (const T*)va, na, kb, (const T*)vb, nb, from_bits<T>(fill_a_bits),                         \
                     from_bits<T>(fill_b_bits), from_bits<O>(fill_out_bits), (const int64_t*)nullptr, ws,       \
                     total_dev, out_keys, (O*)out_vals, total_host)
  switch (val_dtype) {
    case SPAMD_F32: if (to_bool) MP_FUSED(float, uint8_t); else MP_FUSED(float, float); break;
    case SPAMD_F64: if (to_bool) MP_FUSED(double, uint8_t); else MP_FUSED(double, double); break;
    case SPAMD_I32: if (to_bool) MP_FUSED(int32_t, uint8_t); else MP_FUSED(int32_t, int32_t); break;
    case SPAMD_I64: if (to_bool) MP_FUSED(int64_t, uint8_t); else MP_FUSED(int64_t, int64_t); break;
    case SPAMD_U8: MP_FUSED(uint8_t, uint8_t); break;
    default: return SPAMD_ETYPE;
  }
#undef MP_FUSED
  return launch_status();
}
