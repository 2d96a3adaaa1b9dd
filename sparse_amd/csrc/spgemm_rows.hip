// A4 / A5, row-local form: C = A @ B with both operands compressed by rows (reference `_csr_csr_count_nnz` +
// `_dot_csr_csr`, sparse/numba_backend/_common.py:543-570,639-717).
//
// The global expand-sort-compress of spgemm.hip writes every product to HBM and radix-sorts all of them by
// (row, column): 5 passes over 12 bytes per product.  But the products already arrive grouped by output row, so
// only the columns inside a row need sorting — and a row's products fit in LDS for all but the heaviest rows.
// One workgroup per output row:
//   expand   product p of the row -> (A element e, offset inside B row k_e) by a binary search in an LDS prefix
//            array of the B row lengths; key = column, value = a * b (one rounded multiply);
//   order    bucket by the high column bits with LDS atomics, then rank inside the ~5-element buckets (no sort network,
//            no radix sort; see "the row kernel" below);
//   compress every run of equal columns is summed left to right, in the order of A's elements, by its head product
//            (`sums[j] += ...` in the reference's order: bit-identical).
// Rows are written to a scratch area at their product offset (an upper bound of their length); a second kernel
// packs them once the row lengths are scanned.  Rows are served by size class (workgroup size x items per thread);
// a row whose products (or A elements) exceed the largest class makes the caller fall back to spgemm.hip.
#include <string.h>
#include <cstring>
#include <mutex>
#include "common.h"

namespace spamd {

__device__ __forceinline__ int64_t max_seen(const int64_t* p) {  // a stale read only costs a redundant atomic
  return __builtin_nontemporal_load(p);
}

// products per output row: prod[r] = sum over A's elements (r, k) of the length of B row k; maxes = {largest prod, longest A row}.
// A workgroup takes SPG_RP_ROWS consecutive rows and walks their elements FLAT (element e belongs to the last row whose
// pointer is <= e: a binary search over the workgroup's pointers in LDS), four elements per thread and trip so that their
// index loads and then their eight pointer loads are in flight together; a wave whose 64 elements are of one row adds once.
// (Rounds 2-4: a wave per row, one element per lane and trip - 0.49-0.57 ms for config 5's block of 1.25e7 elements.)
constexpr int SPG_RP_ROWS = 64;
template <typename I>
__global__ void __launch_bounds__(256) spgemm_row_products_kernel(int64_t n_row, const I* __restrict__ a_ptr,
                                                                  const I* __restrict__ a_idx, const I* __restrict__ b_ptr,
                                                                  int64_t* __restrict__ prod, int64_t* __restrict__ maxes) {
  __shared__ int64_t rp[SPG_RP_ROWS + 1];
  __shared__ unsigned long long acc[SPG_RP_ROWS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int64_t r0 = (int64_t)blockIdx.x * SPG_RP_ROWS;
  const int nr = (int)(n_row - r0 < SPG_RP_ROWS ? n_row - r0 : SPG_RP_ROWS);
  if (tid <= nr) rp[tid] = (int64_t)a_ptr[r0 + tid];
  if (tid < SPG_RP_ROWS) acc[tid] = 0;
  __syncthreads();
  const int64_t e0 = rp[0], e1 = rp[nr];
  for (int64_t eb = e0 + (tid & ~63) * 4; eb < e1; eb += 256 * 4) {   // (a wave takes 4 x 64 consecutive elements per trip)
    int64_t k[4], lo[4], hi[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t e = eb + 64 * u + lane;
      k[u] = e < e1 ? (int64_t)a_idx[e] : -1;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      lo[u] = k[u] >= 0 ? (int64_t)b_ptr[k[u]] : 0;
      hi[u] = k[u] >= 0 ? (int64_t)b_ptr[k[u] + 1] : 0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t e = eb + 64 * u + lane;
      int t = 0;
#pragma unroll
      for (int step = SPG_RP_ROWS / 2; step >= 1; step >>= 1) t = (t + step <= nr && rp[t + step] <= e) ? t + step : t;
      const unsigned long long len = k[u] >= 0 ? (unsigned long long)(hi[u] - lo[u]) : 0ull;
      const int t0 = __builtin_amdgcn_readfirstlane(t);
      if (__ballot(t != t0) == 0) {          // one row (or nothing but the tail beyond e1, which adds 0)
        unsigned long long v = len;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
        if (lane == 0 && v) atomicAdd(&acc[t0 < nr ? t0 : nr - 1], v);
      } else if (len) {
        atomicAdd(&acc[t < nr ? t : nr - 1], len);
      }
    }
  }
  __syncthreads();
  int64_t m0 = 0, m1 = 0;
  if (tid < nr) {
    m0 = (int64_t)acc[tid];
    m1 = rp[tid + 1] - rp[tid];
    prod[r0 + tid] = m0;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const int64_t y0 = __shfl_xor(m0, d, 64), y1 = __shfl_xor(m1, d, 64);
    m0 = y0 > m0 ? y0 : m0;
    m1 = y1 > m1 ? y1 : m1;
  }
  // (one pair of atomics per workgroup at most: 10^5 same-address atomics from every wave serialise - 2 ms)
  if (tid == 0) {   // (rows sit in the first wave: SPG_RP_ROWS = 64)
    if (m0 > max_seen(maxes)) atomicMax(reinterpret_cast<unsigned long long*>(maxes), (unsigned long long)m0);
    if (m1 > max_seen(maxes + 1)) atomicMax(reinterpret_cast<unsigned long long*>(maxes + 1), (unsigned long long)m1);
  }
}

// ---- the row kernel: bucket + rank --------------------------------------------------------------------------------------
// An output row's P products (P <= N = BLOCK * ITEMS) are ordered by column WITHOUT a sort network or a radix sort:
//   1. bucket   b = floor(column * buckets / n_col) (about one bucket per two products, monotone in the column); every product takes a slot in its bucket
//               with one LDS atomic (arrival order: arbitrary), an exclusive scan of the bucket sizes gives the bucket starts,
//               and (key, value) go to LDS at start[b] + slot, key = (column << EB) | e with e = index of the A element
//               the product comes from.  Products of one column come from distinct A elements (a B row holds a column
//               once), so keys are unique and (column, e) IS the reference's summation order (`sums[j] += ...` over A's
//               elements in order, _common.py:690-705).
//   2. rank     a bucket holds ~2.5 products: ONE thread reads it into registers, orders it by key with a compare-exchange
//               network and sums every run of equal columns left to right (increasing e: bit-identical to the reference);
//               the run heads (column, sum) go back into the bucket's own LDS slots, their number into a head counter.
//   3. emit     an exclusive scan of the head counters gives each bucket's first output slot; buckets are in column order
//               and sorted inside, so rows come out column-sorted and neighbouring threads write neighbouring entries.
// ~6 LDS accesses per product and 7 barriers per row against a 5-pass block radix sort before (13 of 27 ms).
// The scans are hand-written (wave shuffles + one LDS hop); nothing here comes from a library.
// A bucket with more than 16 products (in the end: a column of B that collects many products of one row) makes the
// kernel decline the row (nnz_row = -1): the caller routes it through the global form like the rows above the capacity.
#ifndef SPG_WPE
#define SPG_WPE 4
#endif
constexpr int SPG_BUCKET_MAX = 16;      // products one bucket may hold (registers of the thread that ranks it)

template <int BLOCK>
__device__ __forceinline__ int spg_block_exclusive_scan(int v, int* wsum, int& total) {
  // wsum: BLOCK / 64 + 1 ints of LDS.  Two barriers.
  constexpr int NW = BLOCK / 64;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) wsum[wid] = x;
  __syncthreads();
  if (wid == 0) {
    int w = lane < NW ? wsum[lane] : 0;
    int xs = w;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int y = __shfl_up(xs, d, 64);
      if (lane >= d) xs += y;
    }
    if (lane < NW) wsum[lane] = xs - w;
    if (lane == 63) wsum[NW] = xs;
  }
  __syncthreads();
  total = wsum[NW];
  return x - v + wsum[wid];
}

// exclusive scan of arr[0 .. NB) in place, arr[NB] = total; NB a multiple of BLOCK
template <int BLOCK, int NB>
__device__ __forceinline__ int spg_scan_array(int* arr, int* wsum) {
  constexpr int EPT = NB / BLOCK;
  static_assert(NB % BLOCK == 0 && EPT >= 1, "bucket count is a multiple of the workgroup size");
  int loc[EPT];
  int mine = 0;
#pragma unroll
  for (int j = 0; j < EPT; ++j) {
    loc[j] = arr[threadIdx.x * EPT + j];
    mine += loc[j];
  }
  int total;
  int run = spg_block_exclusive_scan<BLOCK>(mine, wsum, total);
#pragma unroll
  for (int j = 0; j < EPT; ++j) {
    arr[threadIdx.x * EPT + j] = run;
    run += loc[j];
  }
  if (threadIdx.x == 0) arr[NB] = total;
  __syncthreads();
  return total;
}

template <int BLOCK, int ITEMS, int PASSES, typename KEY, typename V>
struct RowRankLayout {
  static constexpr int N = BLOCK * ITEMS;                    // products a row may have
  // products one pass may hold in LDS: with several passes a pass gets 1/PASSES of the products only on average
  // (uniform columns: +- sqrt(N) / 2), so it has room for 9/8 of its share before the row is declined
  static constexpr int CAPP = PASSES == 1 ? N : (N / PASSES) * 9 / 8;
  static constexpr int pow2_at_least(int x) { int p = 1; while (p < x) p <<= 1; return p; }
  // buckets per pass: a power of two, a multiple of BLOCK; with several passes one bucket per product slot (the buckets
  // then hold ~0.7 products: the 5-exchange network instead of the 19-exchange one nearly always)
  static constexpr int NBP = pow2_at_least(PASSES > 2 ? (CAPP > BLOCK ? CAPP : BLOCK) : (CAPP / 2 > BLOCK ? CAPP / 2 : BLOCK));
  static constexpr int STAGE = CAPP < 2048 ? CAPP : 2048;    // A elements staged per chunk
  static constexpr size_t prefix_bytes = ((size_t)STAGE + 2) * sizeof(int) + (size_t)STAGE * (sizeof(int64_t) + sizeof(V)) + 16;
  static constexpr size_t items_bytes = (size_t)CAPP * (sizeof(KEY) + sizeof(V));
  static constexpr size_t region = ((prefix_bytes > items_bytes ? prefix_bytes : items_bytes) + 15) / 16 * 16;
  static constexpr size_t bytes = region + (size_t)(NBP + 1) * sizeof(int) + 16;
};

// rows with lo < products <= hi; V is the value type moved bit-wise except for the multiply / add.
// PASSES > 1: the column range is cut into PASSES equal bucket ranges that go through LDS one after the other (the
// products stay in registers), so that a 15360-product row needs 77 KB of LDS instead of 150 KB and TWO workgroups share
// a CU - the global-load latencies of one row's expansion then overlap the other row's ranking (one 1024-thread
// workgroup per CU measured 31 ms at 10^9 products, exposed latency throughout).
template <int BLOCK, int ITEMS, int PASSES, typename KEY, typename V, typename I>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(BLOCK >= 512 ? SPG_WPE : 1, 8)))
spgemm_rowrank_kernel(int64_t n_col, int col_bits, int ebits, const I* __restrict__ a_ptr, const I* __restrict__ a_idx,
                      const V* __restrict__ a_val, const I* __restrict__ b_ptr, const I* __restrict__ b_idx,
                      const V* __restrict__ b_val, const int64_t* __restrict__ prod_off, int64_t lo, int64_t hi,
                      int* __restrict__ tmp_cols, V* __restrict__ tmp_vals, int64_t* __restrict__ nnz_row) {
#pragma clang fp contract(off)
  using L = RowRankLayout<BLOCK, ITEMS, PASSES, KEY, V>;
  constexpr int NBP = L::NBP, CAPP = L::CAPP;
  extern __shared__ __attribute__((aligned(16))) char raw[];
  __shared__ int wsum[BLOCK / 64 + 2];
  __shared__ int declined;
  const int64_t row = blockIdx.x;
  const int64_t base = prod_off[row];
  const int64_t P64 = prod_off[row + 1] - base;
  if (P64 <= lo || P64 > hi) {
    if (P64 == 0 && lo < 0 && threadIdx.x == 0) nnz_row[row] = 0;
    return;
  }
  const int tid = threadIdx.x;
  const int64_t a0 = (int64_t)a_ptr[row];
  const int nA = (int)((int64_t)a_ptr[row + 1] - a0);  // < 2^ebits (checked by the caller)
  const KEY kmax = ~(KEY)0;

  // ---- stage the A row in LDS: prefix[e] = products of the A elements before e (prefix[nA] = P), the start of B
  // row k_e and the A value.  (Every product then needs only LDS lookups and ONE independent pair of global loads;
  // chasing a_idx -> b_ptr -> b_idx per product serialises three memory latencies per item: 81 ms instead of ~10.)
  // Product p of the row belongs to thread p / ITEMS; its key is (column << ebits) | index of its A element.
  constexpr int STAGE = L::STAGE;
  int* const prefix = reinterpret_cast<int*>(raw);
  int64_t* const bstart = reinterpret_cast<int64_t*>(raw + (size_t)(STAGE + 1) * sizeof(int) + 4);
  V* const aval = reinterpret_cast<V*>(reinterpret_cast<char*>(bstart) + (size_t)STAGE * sizeof(int64_t));
  KEY keys[ITEMS];
  V vals[ITEMS];
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    keys[j] = kmax;  // no product
    vals[j] = V(0);
  }
  if (tid == 0) declined = 0;
  int chunk_done = 0;
  for (int c0 = 0; c0 < nA; c0 += STAGE) {  // chunks of STAGE A elements (one chunk unless the A row is very long)
    const int cn = nA - c0 < STAGE ? nA - c0 : STAGE;
    constexpr int EPT = (STAGE + BLOCK - 1) / BLOCK;
    int len[EPT];
    int mine = 0;
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      const int e = tid * EPT + j;
      len[j] = 0;
      if (e < cn) {
        const int64_t k = (int64_t)a_idx[a0 + c0 + e];
        const int64_t bs = (int64_t)b_ptr[k];
        len[j] = (int)((int64_t)b_ptr[k + 1] - bs);
        bstart[e] = bs;
        aval[e] = a_val[a0 + c0 + e];
      }
      mine += len[j];
    }
    int chunk_total;
    int before = spg_block_exclusive_scan<BLOCK>(mine, wsum, chunk_total);
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      const int e = tid * EPT + j;
      if (e <= cn) prefix[e] = before;
      before += len[j];
    }
    __syncthreads();
    // Product p of the ROW belongs to thread p % BLOCK (item p / BLOCK): for one item index, consecutive lanes read
    // consecutive entries of a B row — a wave-load touches two or three lines instead of 64 (with p / ITEMS = thread, every
    // lane sat in its own line: the expansion alone took 9.7 of the kernel's 18 ms, ablation of 2026-09).  A thread's
    // products are BLOCK apart, so its A element advances by a few steps from one item to the next.
    const int done = chunk_done;  // products of earlier chunks (block-uniform register)
    {
      int e = 0;
      constexpr int G = 8;   // loads in flight per thread (a register budget, not a latency one: 512 threads x 8 x 2)
#pragma unroll
      for (int j0 = 0; j0 < ITEMS; j0 += G) {
        int64_t q[G];
        V av[G];
        int ee[G];
        bool on[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const int p = (j0 + g) * BLOCK + tid - done;
          on[g] = j0 + g < ITEMS && p >= 0 && p < chunk_total;
          q[g] = 0;
          av[g] = V(0);
          ee[g] = 0;
          if (on[g]) {
            while (prefix[e + 1] <= p) ++e;
            q[g] = bstart[e] + (p - prefix[e]);
            av[g] = aval[e];
            ee[g] = c0 + e;
          }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
          if (j0 + g < ITEMS && on[g]) {
            keys[j0 + g] = ((KEY)(unsigned)b_idx[q[g]] << ebits) | (KEY)(unsigned)ee[g];
            vals[j0 + g] = av[g] * b_val[q[g]];
          }
        }
      }
    }
    chunk_done += chunk_total;
    __syncthreads();
  }

#if defined(SPG_ABL) && SPG_ABL == 1   // timing ablation (wrong results): expansion only
  {
    KEY x = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) x ^= keys[j] + (KEY)__builtin_bit_cast(unsigned, (float)vals[j]);
    if (x == (KEY)0x12345) nnz_row[row] = 1;
    if (tid == 0) nnz_row[row] = 0;
    return;
  }
#endif
  KEY* const lkey = reinterpret_cast<KEY*>(raw);
  V* const lval = reinterpret_cast<V*>(raw + (size_t)CAPP * sizeof(KEY));
  int* const cnt = reinterpret_cast<int*>(raw + L::region);
  // global bucket of a column: floor(column * scale / 2^32) with scale = floor(NBP * PASSES * 2^32 / n_col): monotone in the
  // column, < NBP * PASSES, and the passes get equal shares of the COLUMN RANGE (a plain shift of the column would hand pass 0
  // the share 2^k / n_col).  With fewer columns than buckets the columns are spread out the same way (round 6; until then a
  // column was its own bucket there, so pass 0 of an 8-pass class took the first 4096 of, say, 10^4 columns - 41 % of the row's
  // products against the 12.5 % + 1/8 it has room for - and nearly every such row was declined and redone by the global form:
  // float64 10^4 x 10^4 with 100 elements per row 10.0 ms against 2.7 ms in float32, whose classes have two passes).
  const uint64_t scale = (((uint64_t)(NBP * PASSES)) << 32) / (uint64_t)n_col;
  auto bucket_of = [&](KEY key) { return (int)(((uint64_t)(key >> ebits) * scale) >> 32); };
  int row_total = 0;

#pragma unroll 1
  for (int pass = 0; pass < PASSES; ++pass) {
    const int b_lo = pass * NBP;   // this pass ranks the buckets [b_lo, b_lo + NBP)
    // ---- 1. bucket ---------------------------------------------------------------------------------------------------
    for (int i = tid; i <= NBP; i += BLOCK) cnt[i] = 0;
    __syncthreads();   // (also: the staging area / the previous pass's items are dead)
    // a product's slot inside its bucket, 5 bits each, six to a register (31 = not in this pass, or a bucket that
    // overflows SPG_BUCKET_MAX and gets the row declined below): 30 products cost 5 registers instead of 30
    constexpr int SW = (ITEMS + 5) / 6;
    unsigned slots[SW];
#pragma unroll
    for (int w = 0; w < SW; ++w) slots[w] = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      const int b = bucket_of(keys[j]) - b_lo;
      unsigned sl = 31u;
      if (keys[j] != kmax && b >= 0 && b < NBP) {
        const unsigned got = (unsigned)atomicAdd(&cnt[b], 1);
        sl = got < (unsigned)SPG_BUCKET_MAX ? got : 31u;
      }
      slots[j / 6] |= sl << (5 * (j % 6));
    }
    __syncthreads();
    const int in_pass = spg_scan_array<BLOCK, NBP>(cnt, wsum);   // cnt[b] = first LDS position of bucket b
    if (in_pass > CAPP) {   // block-uniform: a skewed column distribution, this pass does not fit LDS
      if (tid == 0) nnz_row[row] = -1;
      return;
    }
#if defined(SPG_ABL) && SPG_ABL == 2   // timing ablation: expansion + bucket counts + scan
    if (pass == PASSES - 1) { if (tid == 0) nnz_row[row] = 0; return; }
    __syncthreads();
    continue;
#endif
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      const unsigned sl = (slots[j / 6] >> (5 * (j % 6))) & 31u;
      if (sl != 31u) {
        const int at = cnt[bucket_of(keys[j]) - b_lo] + (int)sl;
        lkey[at] = keys[j];
        lval[at] = vals[j];
      }
    }
    __syncthreads();

    // ---- 2. rank: one thread per bucket ----------------------------------------------------------------------------------
    // The bucket (~2 products, at most SPG_BUCKET_MAX) is read into registers ONCE, ordered there by (column, A element)
    // with a compare-exchange network, and its runs of equal columns are summed left to right; the heads (column, sum)
    // go back into the bucket's own LDS slots.  No dependent LDS round trips, no divergence beyond the 8 / 16 split.
    constexpr int BPT = NBP / BLOCK;   // consecutive buckets per thread
    // The heads of a thread's buckets are written back COMPACTLY from the thread's first slot on (tb + number of heads so
    // far <= the bucket's own first slot, and the bucket being ranked sits in registers already), so that a thread's
    // heads are one contiguous LDS run [tb, tb + mine): the emission below can then map output slots to LDS slots.
    const int tb = cnt[tid * BPT];
    int mine = 0;
    // (a REAL loop: unrolled eight times with three sorting networks each, the kernel was 400 KB of code - six times the
    // instruction cache two CUs share; nothing in the body is indexed by bb but LDS)
#pragma unroll 1
    for (int bb = 0; bb < BPT; ++bb) {
      const int b = tid * BPT + bb;
      const int s0 = cnt[b];
      const int c = cnt[b + 1] - s0;
      if (c > SPG_BUCKET_MAX) {
        declined = 1;
        continue;
      }
      if (c == 0) continue;
      KEY k[SPG_BUCKET_MAX];
      V v[SPG_BUCKET_MAX];
#pragma unroll
      for (int i = 0; i < SPG_BUCKET_MAX; ++i) {
        const bool on = i < c && (i < 4 || c > 4) && (i < 8 || c > 8);
        k[i] = on ? lkey[s0 + i] : kmax;
        v[i] = on ? lval[s0 + i] : V(0);
      }
#define SPG_CE(x, y)                              \
  {                                               \
    const bool sw = k[y] < k[x];                  \
    const KEY tk = sw ? k[y] : k[x];              \
    k[y] = sw ? k[x] : k[y];                      \
    k[x] = tk;                                    \
    const V tv = sw ? v[y] : v[x];                \
    v[y] = sw ? v[x] : v[y];                      \
    v[x] = tv;                                    \
  }
      if (c > 8) {
        // 16 keys: bitonic sorting network (80 compare-exchanges, all static indices)
#pragma unroll
        for (int size = 2; size <= SPG_BUCKET_MAX; size <<= 1) {
#pragma unroll
          for (int stride = size >> 1; stride > 0; stride >>= 1) {
#pragma unroll
            for (int i = 0; i < SPG_BUCKET_MAX; ++i) {
              const int j = i ^ stride;
              if (j > i) {
                if ((i & size) == 0) SPG_CE(i, j) else SPG_CE(j, i)
              }
            }
          }
        }
      } else if (c > 4) {
        // 8 keys: the 19-exchange network
        SPG_CE(0, 1) SPG_CE(2, 3) SPG_CE(4, 5) SPG_CE(6, 7)
        SPG_CE(0, 2) SPG_CE(1, 3) SPG_CE(4, 6) SPG_CE(5, 7)
        SPG_CE(1, 2) SPG_CE(5, 6) SPG_CE(0, 4) SPG_CE(3, 7)
        SPG_CE(1, 5) SPG_CE(2, 6)
        SPG_CE(1, 4) SPG_CE(3, 6)
        SPG_CE(2, 4) SPG_CE(3, 5)
        SPG_CE(3, 4)
      } else if (c > 1) {
        // 4 keys: five exchanges (the common case: buckets hold less than one product on average)
        SPG_CE(0, 1) SPG_CE(2, 3) SPG_CE(0, 2) SPG_CE(1, 3) SPG_CE(1, 2)
      }
#undef SPG_CE
      // runs of equal columns, summed in increasing A-element order; a run ends where the next column differs
      int h = 0;
      V acc = V(0);
#define SPG_RUN(i)                                                                                         \
  if ((i) < c) {                                                                                           \
    const KEY col = k[(i)] >> ebits;                                                                       \
    const bool first = (i) == 0 || (k[(i) ? (i)-1 : 0] >> ebits) != col;                                   \
    acc = first ? v[(i)] : acc + v[(i)];                                                                   \
    const bool last = (i) + 1 == c || (k[(i) + 1 < SPG_BUCKET_MAX ? (i) + 1 : (i)] >> ebits) != col;       \
    if (last) {                                                                                            \
      lkey[tb + mine + h] = col;                                                                           \
      lval[tb + mine + h] = acc;                                                                           \
      ++h;                                                                                                 \
    }                                                                                                      \
  }
      SPG_RUN(0) SPG_RUN(1) SPG_RUN(2) SPG_RUN(3)
      if (c > 4) {
        SPG_RUN(4) SPG_RUN(5) SPG_RUN(6) SPG_RUN(7)
        if (c > 8) {
          SPG_RUN(8) SPG_RUN(9) SPG_RUN(10) SPG_RUN(11) SPG_RUN(12) SPG_RUN(13) SPG_RUN(14) SPG_RUN(15)
        }
      }
#undef SPG_RUN
      mine += h;
    }
    // ---- 3. emit: the thread's buckets are consecutive, so one scan of the per-thread head counts places them
    int pass_total;
    int at = spg_block_exclusive_scan<BLOCK>(mine, wsum, pass_total);   // (its barriers also publish `declined`)
    if (declined) {   // block-uniform: a bucket (in the end: a column) collects too many products of this row
      if (tid == 0) nnz_row[row] = -1;
      return;
    }
#if defined(SPG_ABL) && SPG_ABL == 3   // timing ablation: everything but the emission
    row_total += pass_total;
    if (PASSES > 1) __syncthreads();
    continue;
#endif
    // Output slot r of this pass lives in LDS slot map[r] (thread t: slots at .. at + mine -> tb .. tb + mine).  The map
    // (16-bit entries) takes the place of the bucket offsets, which are dead now; with it the heads leave in output order,
    // consecutive lanes writing consecutive elements (a thread writing its own ~9 heads one after the other put 64 lanes
    // into 64 different lines per store: 3.3 of the kernel's 18 ms).
    static_assert((size_t)CAPP * sizeof(unsigned short) <= (size_t)(NBP + 1) * sizeof(int) && CAPP <= 65535, "slot map");
    unsigned short* const map = reinterpret_cast<unsigned short*>(cnt);
    // (every thread has read its bucket offsets: the scan above has barriers behind the rank phase)
    for (int i = 0; i < mine; ++i) map[at + i] = (unsigned short)(tb + i);
    __syncthreads();
    {
      const int64_t o = base + row_total;
      for (int r = tid; r < pass_total; r += BLOCK) {
        const int src = map[r];
        tmp_cols[o + r] = (int)lkey[src];
        tmp_vals[o + r] = lval[src];
      }
    }
    row_total += pass_total;
    if (PASSES > 1) __syncthreads();   // the next pass re-uses cnt / lkey / lval, which the emission above still reads
  }
  if (tid == 0) nnz_row[row] = row_total;
}

// pack the rows: out[indptr[row] + i] = tmp[prod_off[row] + i], i < nnz(row); one workgroup per row
template <typename V>
__global__ void __launch_bounds__(256) spgemm_pack_kernel(const int64_t* __restrict__ prod_off,
                                                          const int64_t* __restrict__ out_ptr, const int* __restrict__ tmp_cols,
                                                          const V* __restrict__ tmp_vals, int64_t* __restrict__ out_idx,
                                                          V* __restrict__ out_val, unsigned long long* __restrict__ zeros) {
  const int64_t row = blockIdx.x;
  const int64_t src = prod_off[row], dst = out_ptr[row];
  const int64_t n = out_ptr[row + 1] - dst;
  int nz = 0;   // values whose bits are all zero (+0.0 / 0): what the container's prune would otherwise count in a pass of its own
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    out_idx[dst + i] = (int64_t)tmp_cols[src + i];
    const V v = tmp_vals[src + i];
    out_val[dst + i] = v;
    if constexpr (sizeof(V) == 8) nz += __builtin_bit_cast(uint64_t, v) == 0;
    else if constexpr (sizeof(V) == 4) nz += __builtin_bit_cast(uint32_t, v) == 0;
    else nz += v == V(0);
  }
  if (zeros) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) nz += __shfl_xor(nz, d, 64);
    if ((threadIdx.x & 63) == 0 && nz) atomicAdd(zeros, (unsigned long long)nz);
  }
}

// heavy rows come from the global expand-sort-compress as CSR over ALL rows (src_ptr, int64 columns): put them into
// the scratch at their product offset and record their lengths, so that the pack kernel treats every row alike
template <typename V>
__global__ void __launch_bounds__(256) spgemm_unpack_kernel(const int64_t* __restrict__ rows, const int64_t* __restrict__ src_ptr,
                                                            const int64_t* __restrict__ src_idx, const V* __restrict__ src_val,
                                                            const int64_t* __restrict__ prod_off, int* __restrict__ tmp_cols,
                                                            V* __restrict__ tmp_vals, int64_t* __restrict__ nnz_row) {
  const int64_t row = rows[blockIdx.x];
  const int64_t src = src_ptr[row], dst = prod_off[row];
  const int64_t n = src_ptr[row + 1] - src;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    tmp_cols[dst + i] = (int)src_idx[src + i];
    tmp_vals[dst + i] = src_val[src + i];
  }
  if (threadIdx.x == 0) nnz_row[row] = n;
}

template <int BLOCK, int ITEMS, int PASSES, typename KEY, typename V, typename I>
static int launch_rowrank(int64_t n_row, int64_t n_col, int col_bits, int ebits, const I* a_ptr, const I* a_idx,
                          const V* a_val, const I* b_ptr, const I* b_idx, const V* b_val, const int64_t* prod_off, int64_t lo,
                          int64_t hi, int* tmp_cols, V* tmp_vals, int64_t* nnz_row, hipStream_t s) {
  using L = RowRankLayout<BLOCK, ITEMS, PASSES, KEY, V>;
  static_assert(L::bytes <= 80 * 1024 - 128, "two 512-thread workgroups of a row class share the 160 KB of a CU");
  auto kern = &spgemm_rowrank_kernel<BLOCK, ITEMS, PASSES, KEY, V, I>;
  if (L::bytes > 48 * 1024)
    if (int rc = set_max_dynamic_lds(reinterpret_cast<const void*>(kern), (int)L::bytes)) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)n_row), dim3(BLOCK), L::bytes, s, n_col, col_bits, ebits, a_ptr, a_idx, a_val, b_ptr,
                     b_idx, b_val, prod_off, lo, hi, tmp_cols, tmp_vals, nnz_row);
  return launch_status();
}

// size classes (products per row): 1024, 4096 (one pass through LDS), 8192 and the largest one (two passes over halves of
// the column range; <= 78 KB of LDS each, so two workgroups per CU) for (key width, value width):
//   32-bit keys (column bits + A-element bits <= 32) with 4-byte values: 12288; 64-bit keys with 8-byte values: 7168;
//   every other combination: 8192 - and twice that with 1024 threads and four passes (one workgroup per CU: rare rows)
template <typename KEY, typename V>
struct RowClasses {
  static constexpr int MID = sizeof(KEY) + sizeof(V) <= 12 ? 16 : 14;  // (16-byte items: 14 per thread keep a pass under 80 KB)
  static constexpr int64_t c0 = 256 * 4, c1 = 512 * 8, c2 = 512 * MID;
  static constexpr int TOP = sizeof(KEY) + sizeof(V) <= 8 ? 24 : MID;  // (more products per thread spill: 512 x 30 -> 182 registers)
  static constexpr int64_t c3 = 512 * TOP;
  static constexpr int64_t c4 = 1024 * TOP;   // 1024 threads, eight passes: the same LDS per pass, one workgroup per CU (rare rows)
};

static int spg_bits(int64_t n) {   // smallest b with 2^b > n
  int b = 1;
  while (((int64_t)1 << b) <= n) ++b;
  return b;
}

template <typename KEY, typename V, typename I>
static int rowrank_all(int64_t n_row, int64_t n_col, int col_bits, int ebits, const I* a_ptr, const I* a_idx, const V* a_val,
                       const I* b_ptr, const I* b_idx, const V* b_val, const int64_t* prod_off, int64_t max_prod, int* tmp_cols,
                       V* tmp_vals, int64_t* nnz_row, hipStream_t s) {
  using C = RowClasses<KEY, V>;
#define SPG_CLASS(BLOCK, ITEMS, PASSES, LO, HI)                                                                               \
  {                                                                                                                           \
    int rc = launch_rowrank<BLOCK, ITEMS, PASSES, KEY, V, I>(n_row, n_col, col_bits, ebits, a_ptr, a_idx, a_val, b_ptr, b_idx,  \
                                                             b_val, prod_off, LO, HI, tmp_cols, tmp_vals, nnz_row, s);        \
    if (rc) return rc;                                                                                                        \
  }
  SPG_CLASS(256, 4, 1, -1, C::c0)
  if (max_prod > C::c0) SPG_CLASS(512, 8, 1, C::c0, C::c1)
  if (max_prod > C::c1) SPG_CLASS(512, C::MID, 2, C::c1, C::c2)
  if (max_prod > C::c2 && C::c3 > C::c2) SPG_CLASS(512, C::TOP, 2, C::c2, C::c3)
  if (max_prod > C::c3) SPG_CLASS(1024, C::TOP, 8, C::c3, C::c4)
#undef SPG_CLASS
  return 0;
}

// flags[row] = 1 for the rows the row-local kernel cannot take: more products or A elements than `cap`, or declined by the
// kernel (nnz_row[row] < 0, when nnz_row is given).  counts[0] += such rows, counts[1] += their products, counts[2] += rows
// that are heavy ONLY by the length of their A row.
template <typename I>
__global__ void __launch_bounds__(256) spgemm_classify_kernel(int64_t n_row, const int64_t* __restrict__ prod,
                                                              const I* __restrict__ a_ptr, const int64_t* __restrict__ nnz_row,
                                                              int64_t cap, int64_t* __restrict__ flags,
                                                              unsigned long long* __restrict__ counts) {
  unsigned long long c0 = 0, c1 = 0, c2 = 0;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_row; r += (int64_t)gridDim.x * blockDim.x) {
    const int64_t alen = (int64_t)a_ptr[r + 1] - (int64_t)a_ptr[r];
    const bool by_prod = prod[r] > cap, by_a = alen > cap, by_kernel = nnz_row != nullptr && nnz_row[r] < 0;
    const bool h = by_prod || by_a || by_kernel;
    flags[r] = h ? 1 : 0;
    if (h) {
      ++c0;
      c1 += (unsigned long long)prod[r];
      if (by_a && !by_prod) ++c2;
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    c0 += __shfl_xor(c0, d, 64);
    c1 += __shfl_xor(c1, d, 64);
    c2 += __shfl_xor(c2, d, 64);
  }
  if ((threadIdx.x & 63) == 0 && c0) {
    atomicAdd(counts, c0);
    atomicAdd(counts + 1, c1);
    atomicAdd(counts + 2, c2);
  }
}

}  // namespace spamd

using namespace spamd;

// prod[row] = number of products of output row `row` (prod has n_row + 1 entries, the last one is zeroed: ready for
// spamd_exclusive_scan); maxes[0] = largest prod, maxes[1] = longest A row (device int64[2], zeroed here).
extern "C" int spamd_spgemm_row_products(int idx_dtype, int64_t n_row, const void* a_indptr, const void* a_indices,
                                         const void* b_indptr, int64_t* prod, int64_t* maxes, void* stream) {
  if (n_row < 0) return SPAMD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(maxes, 0, 2 * sizeof(int64_t), s);
  if (e != hipSuccess) return (int)e;
  e = hipMemsetAsync(prod + n_row, 0, sizeof(int64_t), s);
  if (e != hipSuccess) return (int)e;
  if (n_row == 0) return 0;
  SPAMD_DISPATCH_IDX(idx_dtype, I, hipLaunchKernelGGL(spgemm_row_products_kernel<I>, dim3((unsigned)ceil_div(n_row, (int64_t)SPG_RP_ROWS)),
                                                      dim3(256), 0, s, n_row, (const I*)a_indptr, (const I*)a_indices,
                                                      (const I*)b_indptr, prod, maxes))
  return launch_status();
}

// Largest number of products (and of A elements) per row the row-local path takes: depends on the value size and on
// whether (column, A-element index) fits a 32-bit key for this product (n_col columns, A rows of at most max_arow elements).
extern "C" int64_t spamd_spgemm_rows_capacity(int val_dtype, int64_t n_col, int64_t max_arow) {
  const bool k32 = spg_bits(n_col) + spg_bits(max_arow > 0 ? max_arow - 1 : 0) <= 32;
  const bool v4 = val_dtype == SPAMD_F32 || val_dtype == SPAMD_I32;
  if (k32) return v4 ? RowClasses<uint32_t, float>::c4 : RowClasses<uint32_t, double>::c4;
  return v4 ? RowClasses<uint64_t, float>::c4 : RowClasses<uint64_t, double>::c4;
}

// Row-local expand / bucket / rank / emit.  prod_off = exclusive scan of prod (n_row + 1); tmp_cols / tmp_vals hold
// prod_off[n_row] entries; nnz_row[n_row + 1] receives the row lengths of C (last entry untouched), or -1 for a row the
// kernel declined (one of its columns collects more than 64 products: the caller computes such rows with the global form
// and hands them over through spamd_spgemm_unpack, like the rows above the capacity, which are skipped: their nnz_row
// stays as the caller initialised it).  Rows whose A row is longer than the capacity must be left to the global form by
// the caller as well.  max_arow = longest A row (spamd_spgemm_row_products).  n_col < 2^31 - 1.
extern "C" int spamd_spgemm_rows(int val_dtype, int idx_dtype, int64_t n_row, int64_t n_col, const void* a_indptr,
                                 const void* a_indices, const void* a_data, const void* b_indptr, const void* b_indices,
                                 const void* b_data, const int64_t* prod_off, int64_t max_prod, int64_t max_arow,
                                 int* tmp_cols, void* tmp_vals, int64_t* nnz_row, void* stream) {
  if (n_row < 0 || n_col < 0 || n_col >= 2147483647LL || max_arow < 0) return SPAMD_EINVAL;
  if (n_row == 0) return 0;
  const int64_t cap = spamd_spgemm_rows_capacity(val_dtype, n_col, max_arow);
  if (max_prod > cap) max_prod = cap;
  const int col_bits = spg_bits(n_col);                       // the "gone" marker n_col must be representable
  const int ebits = spg_bits(max_arow > 0 ? max_arow - 1 : 0);
  const bool k32 = col_bits + ebits <= 32;
  if (!k32 && col_bits + ebits > 64) return SPAMD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  SPAMD_DISPATCH_VAL(val_dtype, V, {
    SPAMD_DISPATCH_IDX(idx_dtype, I, {
      if (k32)
        return (rowrank_all<uint32_t, V, I>(n_row, n_col, col_bits, ebits, (const I*)a_indptr, (const I*)a_indices,
                                            (const V*)a_data, (const I*)b_indptr, (const I*)b_indices, (const V*)b_data,
                                            prod_off, max_prod, tmp_cols, (V*)tmp_vals, nnz_row, s));
      return (rowrank_all<uint64_t, V, I>(n_row, n_col, col_bits, ebits, (const I*)a_indptr, (const I*)a_indices,
                                          (const V*)a_data, (const I*)b_indptr, (const I*)b_indices, (const V*)b_data, prod_off,
                                          max_prod, tmp_cols, (V*)tmp_vals, nnz_row, s));
    })
  })
  return SPAMD_ETYPE;
}

// Which rows are left to the global form (see spamd_spgemm_rows): flags[n_row + 1] (int64, ready for
// spamd_exclusive_scan + spamd_compact; last entry zeroed) and counts[3] = {rows, their products, rows heavy only by their
// A length} (device int64[3], zeroed here).  nnz_row may be NULL (before the row kernel ran).
extern "C" int spamd_spgemm_classify_rows(int idx_dtype, int64_t n_row, const int64_t* prod, const void* a_indptr,
                                          const int64_t* nnz_row, int64_t cap, int64_t* flags, int64_t* counts, void* stream) {
  if (n_row < 0) return SPAMD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(counts, 0, 3 * sizeof(int64_t), s);
  if (e != hipSuccess) return (int)e;
  e = hipMemsetAsync(flags + n_row, 0, sizeof(int64_t), s);
  if (e != hipSuccess) return (int)e;
  if (n_row == 0) return 0;
  int64_t blocks = ceil_div(n_row, (int64_t)256);
  if (blocks > 4096) blocks = 4096;
  SPAMD_DISPATCH_IDX(idx_dtype, I, hipLaunchKernelGGL(spgemm_classify_kernel<I>, dim3((unsigned)blocks), dim3(256), 0, s, n_row,
                                                      prod, (const I*)a_indptr, nnz_row, cap, flags,
                                                      reinterpret_cast<unsigned long long*>(counts)))
  return launch_status();
}

extern "C" int spamd_spgemm_unpack(int val_dtype, int64_t n_heavy, const int64_t* heavy_rows, const int64_t* src_indptr,
                                   const int64_t* src_indices, const void* src_data, const int64_t* prod_off, int* tmp_cols,
                                   void* tmp_vals, int64_t* nnz_row, void* stream) {
  if (n_heavy < 0) return SPAMD_EINVAL;
  if (n_heavy == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  SPAMD_DISPATCH_VAL(val_dtype, V, hipLaunchKernelGGL(spgemm_unpack_kernel<V>, dim3((unsigned)n_heavy), dim3(256), 0, s, heavy_rows,
                                                      src_indptr, src_indices, (const V*)src_data, prod_off, tmp_cols,
                                                      (V*)tmp_vals, nnz_row))
  return launch_status();
}

extern "C" int spamd_spgemm_pack(int val_dtype, int64_t n_row, const int64_t* prod_off, const int64_t* out_indptr,
                                 const int* tmp_cols, const void* tmp_vals, int64_t* out_indices, void* out_data,
                                 int64_t* zero_count, void* stream) {
  if (n_row < 0) return SPAMD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (zero_count)
    if (hipError_t e = hipMemsetAsync(zero_count, 0, sizeof(int64_t), s); e != hipSuccess) return (int)e;
  if (n_row == 0) return 0;
  SPAMD_DISPATCH_VAL(val_dtype, V, hipLaunchKernelGGL(spgemm_pack_kernel<V>, dim3((unsigned)n_row), dim3(256), 0, s, prod_off,
                                                      out_indptr, tmp_cols, (const V*)tmp_vals, out_indices, (V*)out_data,
                                                      (unsigned long long*)zero_count))
  return launch_status();
}
