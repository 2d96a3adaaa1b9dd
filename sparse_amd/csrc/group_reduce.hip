// A8: grouped reduce over sorted keys in two streaming passes (reference `_reduce_calc` / `_reduce_return`,
// sparse/numba_backend/_coo/core.py:1601-1661 via _sparse_array.py:372-437: `np.ufunc.reduceat` over the runs of
// equal group ids after sorting by the kept axes).
//
// group id = key / divisor, computed on the fly.  Replaces floor_divide + flag_heads + exclusive_scan +
// segment_reduce + compact (five passes, ~8 GB of traffic for 10^8 stored elements) by
//   gr_count_kernel   keys -> number of run heads per 2048-element tile (512 threads x 4)                    (reads 8 B / element)
//   exclusive scan over the tiles (spamd_exclusive_scan's rocPRIM call, a few thousand entries)
//   gr_reduce_kernel  keys + values -> every run that ends inside a tile, plus the tile's open head / tail
//                     partials                                                              (reads 8 B + value)
//   gr_fixup_kernel   one workgroup: chains the open partials across tiles (runs longer than a tile)
// A thread owns 4 consecutive elements and combines them left to right; threads, waves and tiles are combined by
// a segmented scan in a fixed tree order: results are run-to-run reproducible.  Floating-point sums therefore
// differ from the reference's reduceat (pairwise for runs of 8+ elements) only by re-association: parity is to
// 1e-12 relative for f64 sums, exact for integers, max/min and the logical ops.  (rocPRIM's reduce_by_key with a
// transform/zip iterator was tried first: 4.3 ms per 10^8 elements, slower than the five passes.  Folding the count
// pass into the reduce kernel with ticketed tiles + decoupled look-back, as merge.hip does, was also measured: 0.90 ms
// instead of 0.68 ms — a same-address ticket atomic costs ~20 ns and a 2048-element tile is only ~14 ns of work.)
#include <string.h>
#include <cstring>
#include "common.h"
#include <rocprim/rocprim.hpp>

namespace spamd {

constexpr int GR_THREADS = 512;
constexpr int GR_ITEMS = 4;  // 32 bytes of keys per lane, two 16-byte loads (8 items per thread was no faster, 2 items
                             // doubles the segmented-scan work per element)
constexpr int GR_TILE = GR_THREADS * GR_ITEMS;

// k / d for 0 <= k: reciprocal multiply in double, then an exact integer correction (a hardware 64-bit divide
// is ~100 instructions)
struct GroupOf {
  int64_t d;
  double rd;
  __device__ __forceinline__ int64_t operator()(int64_t k) const {
    int64_t q = (int64_t)((double)k * rd);
    int64_t r = k - q * d;
    while (r < 0) { --q; r += d; }
    while (r >= d) { ++q; r -= d; }
    return q;
  }
};

// The same for keys below 2^53 (every array with fewer than 2^53 elements), entirely in double precision: k, the
// quotient and the remainder k - q*d are all exactly representable, so one fma decides the +-1 correction and no
// 64-bit integer multiply or conversion back is needed.  Group ids are compared (and stored) as doubles.
struct GroupOfD {
  double d, rd;
  __device__ __forceinline__ double operator()(int64_t k) const {
    const double kd = (double)k;
    double q = __builtin_floor(kd * rd);
    const double r = __builtin_fma(-q, d, kd);
    if (r < 0.0) q -= 1.0;
    if (r >= d) q += 1.0;
    return q;
  }
};

enum { GR_ADD = 0, GR_MUL, GR_MAX, GR_MIN, GR_OR, GR_AND, GR_FMAX, GR_FMIN };

template <typename T>
__device__ __forceinline__ T gr_apply(int op, T x, T y) {
  switch (op) {
    case GR_ADD: return (T)(x + y);
    case GR_MUL: return (T)(x * y);
    case GR_MAX: return (x != x) ? x : ((y != y) ? y : (x > y ? x : y));  // NaN propagates (np.maximum)
    case GR_MIN: return (x != x) ? x : ((y != y) ? y : (x < y ? x : y));
    case GR_OR: return (T)((x != (T)0) || (y != (T)0));
    case GR_FMAX: return (x != x) ? y : ((y != y) ? x : (x > y ? x : y));  // NaN is skipped (np.fmax)
    case GR_FMIN: return (x != x) ? y : ((y != y) ? x : (x < y ? x : y));
    default: return (T)((x != (T)0) && (y != (T)0));
  }
}

// a partial result: `c` elements combined into `v` (c == 0: empty)
template <typename T, typename C = int64_t>
struct Part {
  T v;
  C c;
};
template <typename T, typename C>
__device__ __forceinline__ Part<T, C> gr_join(int op, Part<T, C> a, Part<T, C> b) {  // a then b (a is to the left)
  if (a.c == 0) return b;
  if (b.c == 0) return a;
  return Part<T, C>{gr_apply(op, a.v, b.v), (C)(a.c + b.c)};
}

// summary of a span for the segmented scan: does it contain a head, and the partial after its last head
// (or of the whole span if it has none)
template <typename T, typename C = int64_t>
struct Span {
  Part<T, C> tail;
  int heads;
};
template <typename T, typename C>
__device__ __forceinline__ Span<T, C> gr_concat(int op, Span<T, C> a, Span<T, C> b) {
  Span<T, C> r;
  r.heads = a.heads + b.heads;
  r.tail = b.heads ? b.tail : gr_join(op, a.tail, b.tail);
  return r;
}

template <typename T, typename C>
__device__ __forceinline__ Span<T, C> gr_shfl_up(Span<T, C> s, int delta) {
  Span<T, C> r;
  r.tail.v = __shfl_up(s.tail.v, delta, 64);
  r.tail.c = __shfl_up(s.tail.c, delta, 64);
  r.heads = __shfl_up(s.heads, delta, 64);
  return r;
}

// Exclusive segmented scan of one Span per thread over a workgroup of NT threads (NT / 64 waves).
// Returns the concatenation of all spans of lower-numbered threads; *total (if non-null, valid in the last
// thread... returned to every thread through LDS) receives the whole workgroup's span.
template <typename T, typename C, int NT>
__device__ __forceinline__ Span<T, C> gr_block_exclusive(int op, Span<T, C> mine, Span<T, C>* lds_wave,
                                                         Span<T, C>* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  Span<T, C> inc = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    Span<T, C> o = gr_shfl_up(inc, d);
    if (lane >= d) inc = gr_concat(op, o, inc);
  }
  if (lane == 63) lds_wave[wave] = inc;
  __syncthreads();
  Span<T, C> before;  // spans of the earlier waves
  before.tail.c = 0;
  before.tail.v = (T)0;
  before.heads = 0;
  for (int w = 0; w < wave; ++w) before = gr_concat(op, before, lds_wave[w]);
  if (total) {
    Span<T, C> all = before;
    for (int w = wave; w < NT / 64; ++w) all = gr_concat(op, all, lds_wave[w]);
    *total = all;
  }
  Span<T, C> exc = gr_shfl_up(inc, 1);
  if (lane == 0) {
    exc.tail.c = 0;
    exc.tail.v = (T)0;
    exc.heads = 0;
  }
  __syncthreads();
  return gr_concat(op, before, exc);
}

// GR_ITEMS consecutive elements of a thread: 16-byte loads when they are all in range (base is a multiple of 4, the
// arrays are 16-byte aligned: checked by the caller), else element by element (the last thread of the array)
template <typename E>
__device__ __forceinline__ void gr_load(const E* __restrict__ p, int64_t base, int64_t n, E (&x)[GR_ITEMS]) {
  constexpr int PER = sizeof(E) >= 16 ? 1 : (int)(16 / sizeof(E));     // elements per 16-byte load
  constexpr int W = PER < GR_ITEMS ? PER : GR_ITEMS;
  if (base + GR_ITEMS <= n) {
#pragma unroll
    for (int j = 0; j < GR_ITEMS; j += W) {
      const Vec<E, W> v = *reinterpret_cast<const Vec<E, W>*>(p + base + j);
#pragma unroll
      for (int e = 0; e < W; ++e) x[j + e] = v.v[e];
    }
  } else {
#pragma unroll
    for (int j = 0; j < GR_ITEMS; ++j) x[j] = base + j < n ? p[base + j] : (E)0;
  }
}

template <typename G>
__global__ void __launch_bounds__(GR_THREADS) gr_count_kernel(const int64_t* __restrict__ keys, int64_t n, G gof,
                                                              int64_t* __restrict__ tile_heads, int* __restrict__ need_chain,
                                                              int64_t* __restrict__ n_groups) {
  using GT = decltype(gof((int64_t)0));
  __shared__ int wsum[GR_THREADS / 64];
  // (the two words later kernels of the call accumulate into are cleared here instead of by two memset launches: at
  // config-1 sizes every launch is ~4 us of a ~90 us reduction)
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *need_chain = 0;
    n_groups[1] = 0;
  }
  const int64_t base = (int64_t)blockIdx.x * GR_TILE + (int64_t)threadIdx.x * GR_ITEMS;
  int cnt = 0;
  if (base < n) {
    int64_t k[GR_ITEMS];
    gr_load(keys, base, n, k);
    // the group of the element before mine: lane - 1 has it (its last key), except in lane 0 of a wave
    const int64_t left = __shfl_up(k[GR_ITEMS - 1], 1, 64);
    GT prev = (threadIdx.x & 63) ? gof(left) : (base > 0 ? gof(keys[base - 1]) : (GT)-1);
#pragma unroll
    for (int j = 0; j < GR_ITEMS; ++j) {
      if (base + j < n) {
        const GT g = gof(k[j]);
        // (the array's first element starts a run WHATEVER its group: a key of group -1 - not a valid key, but what an
        // unwritten buffer may hold when the slab merge in front declined its ranges and the caller has not read that yet -
        // equals the "nothing before" mark, the array then had no run at all and the fix-up stored at run -1: a device fault,
        // found by tools/fuzz_dense.py in round 6)
        cnt += (g != prev) || (base + j == 0);
        prev = g;
      }
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < GR_THREADS / 64; ++w) t += wsum[w];
    tile_heads[blockIdx.x] = t;
  }
}

// tile_first[b] = index of the first run that STARTS in tile b (exclusive scan of tile_heads).
// Writes gids / vals / counts of every run that starts in this tile and is closed inside it, and
//   open_head[b]  = partial of the elements before the tile's first head (they belong to run tile_first[b]-1)
//   open_tail[b]  = partial after the tile's last head (run tile_first[b+1]-1), or of the whole tile if it has no head
template <typename T, typename G>
__global__ void __launch_bounds__(GR_THREADS)
gr_reduce_kernel(int op, const int64_t* __restrict__ keys, const T* __restrict__ data, int64_t n, G gof,
                 const int64_t* __restrict__ tile_first, int64_t* __restrict__ gids, T* __restrict__ vals,
                 int64_t* __restrict__ counts, Part<T>* __restrict__ open_head, Part<T>* __restrict__ open_tail) {
  using GT = decltype(gof((int64_t)0));
  using C = int;  // run lengths inside a tile fit 32 bits (one shuffle less per scan step)
  __shared__ Span<T, C> lds_wave[GR_THREADS / 64];
  const int64_t base = (int64_t)blockIdx.x * GR_TILE + (int64_t)threadIdx.x * GR_ITEMS;
  GT g[GR_ITEMS];
  T v[GR_ITEMS];
  bool h[GR_ITEMS];
  int64_t kk[GR_ITEMS];
#pragma unroll
  for (int j = 0; j < GR_ITEMS; ++j) {
    kk[j] = 0;
    v[j] = (T)0;
  }
  if (base < n) {
    gr_load(keys, base, n, kk);
    gr_load(data, base, n, v);
  }
  // (every lane takes part in the shuffle; lanes past the end hold zeros nobody reads)
  const int64_t left = __shfl_up(kk[GR_ITEMS - 1], 1, 64);
  GT prev = (GT)-1;
  if (base < n) prev = (threadIdx.x & 63) ? gof(left) : (base > 0 ? gof(keys[base - 1]) : (GT)-1);
  Span<T, C> mine;
  mine.heads = 0;
  mine.tail.c = 0;
  mine.tail.v = (T)0;
#pragma unroll
  for (int j = 0; j < GR_ITEMS; ++j) {
    h[j] = false;
    if (base + j < n) {
      g[j] = gof(kk[j]);
      h[j] = (g[j] != prev) || (base + j == 0);       // (as in gr_count_kernel: element 0 is a head whatever its group)
      prev = g[j];
      if (h[j]) {
        mine.heads += 1;
        mine.tail = Part<T, C>{v[j], 1};
      } else {
        mine.tail = gr_join(op, mine.tail, Part<T, C>{v[j], 1});
      }
    }
  }
  Span<T, C> all;
  const Span<T, C> before = gr_block_exclusive<T, C, GR_THREADS>(op, mine, lds_wave, &all);
  // second walk: close runs
  const int64_t first = tile_first[blockIdx.x];
  int64_t k = before.heads;      // heads seen so far in this tile
  Part<T, C> run = before.tail;  // the open run entering my items
#pragma unroll
  for (int j = 0; j < GR_ITEMS; ++j) {
    if (base + j < n) {
      if (h[j]) {
        if (k == 0) {
          open_head[blockIdx.x] = Part<T>{run.v, run.c};  // elements of a run that started in an earlier tile
        } else {
          vals[first + k - 1] = run.v;
          counts[first + k - 1] = run.c;
        }
        gids[first + k] = (int64_t)g[j];
        ++k;
        run = Part<T, C>{v[j], 1};
      } else {
        run = gr_join(op, run, Part<T, C>{v[j], 1});
      }
    }
  }
  if (threadIdx.x == GR_THREADS - 1) {
    open_tail[blockIdx.x] = Part<T>{all.tail.v, all.tail.c};
    if (all.heads == 0) open_head[blockIdx.x] = Part<T>{(T)0, 0};  // everything is in open_tail
  }
}

// Closing the runs that are open at a tile boundary.  Run tile_first[b] - 1 is open when tile b begins; if tile b
// has a head it ends there and its value is (open tails since its head) + open_head[b]; the run open at the very
// end (total - 1) ends at the virtual tile `ntiles`.
// Fast path, one thread per tile: walk back over at most GR_WALK head-less tiles.  A longer stretch (a run
// spanning more than ~64k elements) raises *need_chain and the single-workgroup segmented scan below redoes all
// of them.
constexpr int GR_WALK = 32;

template <typename T>
__global__ void __launch_bounds__(256)
gr_fix_fast_kernel(int op, int64_t ntiles, const int64_t* __restrict__ tile_heads, const int64_t* __restrict__ tile_first,
                   const Part<T>* __restrict__ open_head, const Part<T>* __restrict__ open_tail, T* __restrict__ vals,
                   int64_t* __restrict__ counts, int64_t* __restrict__ n_groups, int* __restrict__ need_chain) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;  // 1 .. ntiles
  if (b > ntiles) return;
  if (b == ntiles) *n_groups = tile_first[ntiles];
  if (b < ntiles && tile_heads[b] == 0) return;
  int64_t s = b - 1;  // first tile of the head-less stretch that precedes b (s == b: none)
  int steps = 0;
  while (s >= 0 && tile_heads[s] == 0) {
    if (++steps > GR_WALK) {
      *need_chain = 1;
      return;
    }
    --s;
  }
  // s = last tile before b that has a head (or -1): the run consists of open_tail[s], the whole tiles s+1 .. b-1
  // (their open_tail), and open_head[b]
  Part<T> run{(T)0, 0};
  for (int64_t t = s < 0 ? 0 : s; t < b; ++t) run = gr_join(op, run, open_tail[t]);
  if (b < ntiles) run = gr_join(op, run, open_head[b]);
  const int64_t r = tile_first[b] - 1;
  if (run.c) {
    vals[r] = run.v;
    counts[r] = run.c;
  }
}

// One workgroup: the same by a segmented scan over all tiles, for arbitrarily long runs.
template <typename T>
__device__ __forceinline__ void gr_fixup_body(int op, int64_t ntiles, const int64_t* __restrict__ tile_heads,
                                              const int64_t* __restrict__ tile_first, const Part<T>* __restrict__ open_head,
                                              const Part<T>* __restrict__ open_tail, T* __restrict__ vals,
                                              int64_t* __restrict__ counts, int64_t* __restrict__ n_groups, Span<T>* lds_wave) {
  const int64_t per = (ntiles + 1023) / 1024;
  const int64_t b0 = (int64_t)threadIdx.x * per, b1 = b0 + per < ntiles ? b0 + per : ntiles;
  Span<T> mine;
  mine.heads = 0;
  mine.tail.c = 0;
  mine.tail.v = (T)0;
  for (int64_t b = b0; b < b1; ++b) {
    Span<T> s;
    s.heads = tile_heads[b] ? 1 : 0;
    s.tail = open_tail[b];
    mine = gr_concat(op, mine, s);
  }
  const Span<T> before = gr_block_exclusive<T, int64_t, 1024>(op, mine, lds_wave, nullptr);
  Part<T> carry = before.tail;  // the open run entering tile b0
  for (int64_t b = b0; b < b1; ++b) {
    if (tile_heads[b]) {
      const Part<T> done = gr_join(op, carry, open_head[b]);
      if (done.c) {
        vals[tile_first[b] - 1] = done.v;
        counts[tile_first[b] - 1] = done.c;
      }
      carry = open_tail[b];
    } else {
      carry = gr_join(op, carry, open_tail[b]);
    }
    if (b == ntiles - 1) {
      const int64_t total = tile_first[ntiles];
      if (carry.c) {
        vals[total - 1] = carry.v;
        counts[total - 1] = carry.c;
      }
      *n_groups = total;
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(1024)
gr_fixup_kernel(int op, int64_t ntiles, const int64_t* __restrict__ tile_heads, const int64_t* __restrict__ tile_first,
                const Part<T>* __restrict__ open_head, const Part<T>* __restrict__ open_tail, T* __restrict__ vals,
                int64_t* __restrict__ counts, int64_t* __restrict__ n_groups, const int* __restrict__ need_chain) {
  __shared__ Span<T> lds_wave[1024 / 64];
  if (*need_chain == 0) return;
  gr_fixup_body<T>(op, ntiles, tile_heads, tile_first, open_head, open_tail, vals, counts, n_groups, lds_wave);
}

// Both of the above in ONE launch of one workgroup, for up to GR_FIX_ONE_TILES tiles (round 6: at launch-bound sizes the two
// launches cost 5 + 4 us for a few hundred tiles; the chained form's launch did nothing but read `need_chain`).  The fast
// part asks for everything a tile usually needs - its own and its left neighbour's head counts, the neighbour's open tail,
// its own open head and first run - before looking at any of it (one memory round trip instead of five one behind the
// other), and walks further back only when the neighbour has no head.  Same joins in the same order as the two kernels.
#ifndef SPAMD_GR_FIX_ONE_TILES
#define SPAMD_GR_FIX_ONE_TILES 16384
#endif
constexpr int64_t GR_FIX_ONE_TILES = SPAMD_GR_FIX_ONE_TILES;

template <typename T>
__global__ void __launch_bounds__(1024)
gr_fix_one_kernel(int op, int64_t ntiles, const int64_t* __restrict__ tile_heads, const int64_t* __restrict__ tile_first,
                  const Part<T>* __restrict__ open_head, const Part<T>* __restrict__ open_tail, T* __restrict__ vals,
                  int64_t* __restrict__ counts, int64_t* __restrict__ n_groups) {
  __shared__ Span<T> lds_wave[1024 / 64];
  __shared__ int chain;
  if (threadIdx.x == 0) chain = 0;
  __syncthreads();
  for (int64_t b = (int64_t)threadIdx.x + 1; b <= ntiles; b += 1024) {
    const int64_t bc = b < ntiles ? b : ntiles - 1;       // (clamped: every load below is unconditional)
    const int64_t hb = tile_heads[bc], hp = tile_heads[b - 1], tf = tile_first[b];
    const Part<T> tp = open_tail[b - 1], oh = open_head[bc];
    if (b == ntiles) *n_groups = tf;
    if (b < ntiles && hb == 0) continue;
    Part<T> run{(T)0, 0};
    if (hp != 0 || b == 1) {
      run = gr_join(op, run, tp);
    } else {
      int64_t s = b - 2;  // first tile of the head-less stretch that precedes b
      int steps = 1;
      bool far = false;
      while (s >= 0 && tile_heads[s] == 0) {
        if (++steps > GR_WALK) {
          far = true;
          break;
        }
        --s;
      }
      if (far) {
        chain = 1;
        continue;
      }
      for (int64_t t = s < 0 ? 0 : s; t < b; ++t) run = gr_join(op, run, open_tail[t]);
    }
    if (b < ntiles) run = gr_join(op, run, oh);
    if (run.c) {
      vals[tf - 1] = run.v;
      counts[tf - 1] = run.c;
    }
  }
  __syncthreads();
  if (chain) gr_fixup_body<T>(op, ntiles, tile_heads, tile_first, open_head, open_tail, vals, counts, n_groups, lds_wave);
}

static int64_t gr_tiles(int64_t n) { return ceil_div(n, (int64_t)GR_TILE); }
static size_t gr_align(size_t x) { return (x + 255) & ~(size_t)255; }

template <typename T>
static int group_reduce_t(int op, int64_t n, const int64_t* keys, int64_t divisor, int64_t key_bound, const T* data,
                          int64_t* gids, T* vals, int64_t* counts, int64_t* n_groups, char* ws, size_t ws_bytes,
                          hipStream_t s) {
  const int64_t nt = gr_tiles(n);
  // workspace: tile_heads[nt+1] | tile_first[nt+1] | open_head[nt] | open_tail[nt] | scan temp
  int64_t* tile_heads = reinterpret_cast<int64_t*>(ws);
  int64_t* tile_first = reinterpret_cast<int64_t*>(ws + gr_align((nt + 1) * 8));
  Part<T>* open_head = reinterpret_cast<Part<T>*>(ws + 2 * gr_align((nt + 1) * 8));
  Part<T>* open_tail = reinterpret_cast<Part<T>*>(ws + 2 * gr_align((nt + 1) * 8) + gr_align(nt * sizeof(Part<T>)));
  int* need_chain = reinterpret_cast<int*>(ws + 2 * gr_align((nt + 1) * 8) + 2 * gr_align(nt * sizeof(Part<T>)));
  char* scan_ws = reinterpret_cast<char*>(need_chain) + 256;
  size_t scan_bytes = ws_bytes - (size_t)(scan_ws - ws);
  // keys are < key_bound (the caller's array size): below 2^53 the group ids are computed in double precision
  const bool small = key_bound > 0 && key_bound <= ((int64_t)1 << 53);
  const GroupOf gof{divisor, 1.0 / (double)divisor};
  const GroupOfD gofd{(double)divisor, 1.0 / (double)divisor};
  if (small)
    hipLaunchKernelGGL(gr_count_kernel<GroupOfD>, dim3((unsigned)nt), dim3(GR_THREADS), 0, s, keys, n, gofd, tile_heads, need_chain,
                       n_groups);
  else
    hipLaunchKernelGGL(gr_count_kernel<GroupOf>, dim3((unsigned)nt), dim3(GR_THREADS), 0, s, keys, n, gof, tile_heads, need_chain,
                       n_groups);
  if (nt + 1 <= SMALL_SCAN_MAX) {
    hipLaunchKernelGGL(small_exclusive_scan_kernel, dim3(1), dim3(1024), 0, s, tile_heads, tile_first, (int)(nt + 1));
  } else {
    hipError_t e = rocprim::exclusive_scan(scan_ws, scan_bytes, tile_heads, tile_first, (int64_t)0, (size_t)(nt + 1),
                                           rocprim::plus<int64_t>(), s);
    if (e != hipSuccess) return (int)e;
  }
  if (small)
    hipLaunchKernelGGL((gr_reduce_kernel<T, GroupOfD>), dim3((unsigned)nt), dim3(GR_THREADS), 0, s, op, keys, data, n, gofd,
                       tile_first, gids, vals, counts, open_head, open_tail);
  else
    hipLaunchKernelGGL((gr_reduce_kernel<T, GroupOf>), dim3((unsigned)nt), dim3(GR_THREADS), 0, s, op, keys, data, n, gof,
                       tile_first, gids, vals, counts, open_head, open_tail);
  if (nt <= GR_FIX_ONE_TILES) {
    hipLaunchKernelGGL(gr_fix_one_kernel<T>, dim3(1), dim3(1024), 0, s, op, nt, tile_heads, tile_first, open_head, open_tail,
                       vals, counts, n_groups);
    return launch_status();
  }
  hipLaunchKernelGGL(gr_fix_fast_kernel<T>, dim3((unsigned)ceil_div(nt, (int64_t)256)), dim3(256), 0, s, op, nt, tile_heads,
                     tile_first, open_head, open_tail, vals, counts, n_groups, need_chain);
  hipLaunchKernelGGL(gr_fixup_kernel<T>, dim3(1), dim3(1024), 0, s, op, nt, tile_heads, tile_first, open_head, open_tail,
                     vals, counts, n_groups, need_chain);
  return launch_status();
}


// ---- everything reduced (`x.sum()`, `x.max()` ... with axis=None): one group, so the keys are not read at all --------------
// The grouped path above reads the keys twice (count + reduce: 16 of its 16 + 8 bytes per f64 element) to find run heads
// that cannot exist.  Here a workgroup folds one contiguous piece of the values (16-byte loads, four in flight per lane;
// a lane folds its own elements left to right, lanes / waves / pieces are joined in index order: reproducible), and the
// last workgroup to finish (a ticket) joins the pieces and writes the result in group_reduce's output form: one group
// with id 0, its value and its count.
constexpr int RA_THREADS = 512;
constexpr int RA_MAX_PIECES = 2048;     // what the workspace holds
constexpr int RA_PIECES = 256;
#ifndef SPAMD_RA_U
#define SPAMD_RA_U 4
#endif
constexpr int RA_U = SPAMD_RA_U;          // 16-byte loads in flight per lane

template <typename T>
__device__ __forceinline__ Part<T, int> ra_block_join(int op, Part<T, int> mine, Part<T, int>* lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {       // lanes in index order: lane l ends up holding l-d+1 .. l
    Part<T, int> o;
    o.v = __shfl_up(mine.v, d, 64);
    o.c = __shfl_up(mine.c, d, 64);
    if (lane >= d) mine = gr_join(op, o, mine);
  }
  if (lane == 63) lds[wave] = mine;
  __syncthreads();
  Part<T, int> all = lds[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) all = gr_join(op, all, lds[w]);
  __syncthreads();
  return all;
}

template <typename T, bool VEC>
__global__ void __launch_bounds__(RA_THREADS)
ra_reduce_kernel(int op, const T* __restrict__ data, int64_t n, int64_t piece, T* part_v,
                 int* part_c, unsigned* ticket, int64_t* __restrict__ group_ids,
                 T* __restrict__ values, int64_t* __restrict__ counts, int64_t* __restrict__ n_groups) {
  constexpr int V = VEC ? (int)(16 / sizeof(T)) : 1;
  __shared__ Part<T, int> lds[RA_THREADS / 64];
  __shared__ unsigned last;
  const int64_t lo = (int64_t)blockIdx.x * piece, hi = lo + piece < n ? lo + piece : n;
  Part<T, int> acc{(T)0, 0};
  // a lane's elements: vectors threadIdx.x, threadIdx.x + 512, ... of the piece - NOT index order across lanes, which is fine for the
  // ops here only because each is associative and commutative up to floating-point re-association (the order is still fixed)
  int64_t at = lo + (int64_t)threadIdx.x * V;
  constexpr int64_t STEP = (int64_t)RA_THREADS * V;
  if constexpr (VEC) {
    for (; at + (RA_U - 1) * STEP + V <= hi; at += RA_U * STEP) {
      Vec<T, V> v[RA_U];
#pragma unroll
      for (int u = 0; u < RA_U; ++u) v[u] = *reinterpret_cast<const Vec<T, V>*>(data + at + u * STEP);
#pragma unroll
      for (int u = 0; u < RA_U; ++u)
#pragma unroll
        for (int e = 0; e < V; ++e) acc = gr_join(op, acc, Part<T, int>{v[u].v[e], 1});
    }
    for (; at + V <= hi; at += STEP) {
      const Vec<T, V> v = *reinterpret_cast<const Vec<T, V>*>(data + at);
#pragma unroll
      for (int e = 0; e < V; ++e) acc = gr_join(op, acc, Part<T, int>{v.v[e], 1});
    }
  }
  if (at < hi) {                                      // the array's last, partial vector (or every element when not VEC)
    if constexpr (VEC) {
      for (int64_t i = at; i < hi; ++i) acc = gr_join(op, acc, Part<T, int>{data[i], 1});
    } else {
      for (; at < hi; at += STEP) acc = gr_join(op, acc, Part<T, int>{data[at], 1});
    }
  }
  Part<T, int> all = ra_block_join(op, acc, lds);
  if (gridDim.x > 1) {
    if (threadIdx.x == 0) {
      part_v[blockIdx.x] = all.v;
      part_c[blockIdx.x] = all.c ? 1 : 0;
      __threadfence();
      last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    acc = Part<T, int>{(T)0, 0};
    const int per = (int)((gridDim.x + RA_THREADS - 1) / RA_THREADS);        // consecutive pieces per lane: index order
    for (int j = 0; j < per; ++j) {
      const int i = (int)threadIdx.x * per + j;
      if (i < (int)gridDim.x) {
        Part<T, int> o{part_v[i], part_c[i]};
        acc = gr_join(op, acc, o);
      }
    }
    all = ra_block_join(op, acc, lds);
    if (threadIdx.x == 0) *ticket = 0;               // the word is ready for the next call on this workspace
  }
  if (threadIdx.x == 0) {
    group_ids[0] = 0;
    values[0] = all.v;
    counts[0] = n;
    n_groups[0] = 1;
    n_groups[1] = 0;
  }
}

template <typename T>
static int reduce_all_t(int op, int64_t n, const T* data, int64_t* group_ids, T* values, int64_t* counts, int64_t* n_groups,
                        char* ws, hipStream_t s) {
  const bool vec = ((uintptr_t)data % 16) == 0;
  const int64_t unit = (int64_t)RA_THREADS * (16 / (int64_t)sizeof(T)) * RA_U;     // a piece is whole rounds of RA_U vectors per lane
  int64_t pieces = ceil_div(n, unit);
  // one workgroup per CU: every piece ends with an atomic on ONE ticket word (~20 ns each, serialised).  10^8 f64 elements:
  // 128 pieces 229 us, 256 142 us (5.6 TB/s), 512 154 us, 1024 176 us, 2048 224 us; 2 / 4 / 8 loads in flight at 256
  // pieces: 174 / 142 / 138 us (tools/r06/reduce_all_ab.sh)
  if (pieces > RA_PIECES) pieces = RA_PIECES;
  const int64_t piece = ceil_div(ceil_div(n, pieces), unit) * unit;
  pieces = ceil_div(n, piece);
  unsigned* ticket = (unsigned*)ws;
  T* part_v = (T*)(ws + 256);
  int* part_c = (int*)(ws + 256 + gr_align(RA_MAX_PIECES * sizeof(T)));
  if (vec)
    hipLaunchKernelGGL((ra_reduce_kernel<T, true>), dim3((unsigned)pieces), dim3(RA_THREADS), 0, s, op, data, n, piece, part_v,
                       part_c, ticket, group_ids, values, counts, n_groups);
  else
    hipLaunchKernelGGL((ra_reduce_kernel<T, false>), dim3((unsigned)pieces), dim3(RA_THREADS), 0, s, op, data, n, piece, part_v,
                       part_c, ticket, group_ids, values, counts, n_groups);
  return (int)hipGetLastError();
}

}  // namespace spamd

using namespace spamd;

extern "C" int64_t spamd_group_reduce_ws_bytes(int val_dtype, int64_t n) {
  const int64_t nt = gr_tiles(n > 0 ? n : 1);
  size_t scan = 0;
  int64_t* p = nullptr;
  hipError_t e = rocprim::exclusive_scan(nullptr, scan, p, p, (int64_t)0, (size_t)(nt + 1), rocprim::plus<int64_t>(),
                                         (hipStream_t)0);
  if (e != hipSuccess) return -(int64_t)e;
  return (int64_t)(2 * gr_align((nt + 1) * 8) + 2 * gr_align(nt * 16) + 256 + scan + 256);
}

extern "C" int spamd_group_reduce(int op, int val_dtype, int64_t n, const int64_t* keys, int64_t divisor,
                                  int64_t key_bound, const void* data, int64_t* group_ids, void* values, int64_t* counts,
                                  int64_t* n_groups, void* ws, int64_t ws_bytes, void* stream) {
  if (n < 0 || divisor <= 0 || op < 0 || op > GR_FMIN) return SPAMD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) return (int)hipMemsetAsync(n_groups, 0, sizeof(int64_t), s);
  if (ws_bytes < spamd_group_reduce_ws_bytes(val_dtype, n) || ((uintptr_t)ws % 16) || ((uintptr_t)keys % 16) || ((uintptr_t)data % 16))
    return SPAMD_EINVAL;
  // the tile_heads array has nt + 1 entries for the scan; the last one is ignored by it but must be readable
#define GR_CASE(CODE, T)                                                                                           \
  case CODE:                                                                                                       \
    return group_reduce_t<T>(op, n, keys, divisor, key_bound, (const T*)data, group_ids, (T*)values, counts, n_groups, (char*)ws, \
                             (size_t)ws_bytes, s);
  switch (val_dtype) {
    GR_CASE(SPAMD_F32, float)
    GR_CASE(SPAMD_F64, double)
    GR_CASE(SPAMD_I32, int32_t)
    GR_CASE(SPAMD_I64, int64_t)
    GR_CASE(SPAMD_U8, uint8_t)
    default: return SPAMD_ETYPE;
  }
#undef GR_CASE
}

extern "C" int64_t spamd_reduce_all_ws_bytes(void) {
  return (int64_t)(256 + gr_align(RA_MAX_PIECES * 8) + gr_align(RA_MAX_PIECES * 4));
}

extern "C" int spamd_reduce_all(int op, int val_dtype, int64_t n, const void* data, int64_t* group_ids, void* values,
                                int64_t* counts, int64_t* n_groups, void* ws, int64_t ws_bytes, void* stream) {
  if (n < 0 || op < 0 || op > GR_FMIN) return SPAMD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) return (int)hipMemsetAsync(n_groups, 0, sizeof(int64_t), s);
  if (ws_bytes < spamd_reduce_all_ws_bytes() || ((uintptr_t)ws % 16)) return SPAMD_EINVAL;
#define RA_CASE(CODE, T) \
  case CODE: return reduce_all_t<T>(op, n, (const T*)data, group_ids, (T*)values, counts, n_groups, (char*)ws, s);
  switch (val_dtype) {
    RA_CASE(SPAMD_F32, float)
    RA_CASE(SPAMD_F64, double)
    RA_CASE(SPAMD_I32, int32_t)
    RA_CASE(SPAMD_I64, int64_t)
    RA_CASE(SPAMD_U8, uint8_t)
    default: return SPAMD_ETYPE;
  }
#undef RA_CASE
}
