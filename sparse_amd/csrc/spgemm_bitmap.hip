// A4 / A5, row-local form for WIDE rows of a matrix with at most 2^20 columns (round 4): C = A @ B with both operands
// compressed by rows (reference `_csr_csr_count_nnz` + `_dot_csr_csr`, sparse/numba_backend/_common.py:543-570,639-717).
//
// spgemm_rows.hip orders a row's products with LDS bucket counts + per-bucket sorting networks, writes the row to a
// scratch area at its product offset and needs a pack kernel once every row length is known: ~20 barrier-separated phases
// per row, two passes over the column range, 69 GB of traffic for 20 GB of algorithmic bytes at BASELINE config 5
// (28.8 ms for one GPU's share).  Here the column ORDER comes from a bitmap instead of a sort, and rows go straight to
// their final place (13.1 ms, bit-identical):
//   * one persistent 1024-thread workgroup per CU keeps a bitmap of the output row's columns in LDS (n_col bits: 125 KB
//     at config 5).  Every product sets its column's bit with one returning LDS atomic; the position of a column in the
//     sorted row is the number of set bits below it (a popcount scan over the bitmap + three LDS reads per product), so
//     nothing is sorted and nothing is ranked against anything else;
//   * once every position is known the bitmap has served, and its LDS space takes the row itself: the first product to
//     have set a bit writes (column, value) at its position.  Products that found their bit already set (two products of
//     one output element: ~50 of 10^4 at config 5) are parked in a small LDS list together with the first arriver of
//     their column (which recognises itself through a 16384-bit filter); a wave per entry scans the list, and the entry with
//     the smallest A-element index of each column adds the column's products left to right in A's order - the reference's
//     `sums[j] += ...` order (`_common.py:690-705`), bit-identical to spgemm_rows.hip and to the global form;
//   * the row length is known right after the popcount scan, BEFORE anything is emitted: rows are dealt round-robin to the
//     workgroups, a decoupled look-back over one state word per row gives the row's offset in the result, and the row
//     leaves LDS once, with coalesced stores, to its final place.  No scratch rows, no pack kernel, no scan over the row
//     lengths; the exact zeros written are counted on the way (the result container's prune then has nothing to read);
//   * a row's products are prefetched: while row r is finished and written, the loads of row r+1's products are in flight
//     (into registers) and row r+2's A elements are being fetched.  Barriers wait for LDS only (`s_waitcnt lgkmcnt(0)` +
//     `s_barrier`): `__syncthreads()` would drain the prefetch at every phase.
// Measured on the way (config-5 share, ms per product through `a @ b`; bucket kernels 28.8): first arrivers storing
// straight to HBM 107 (scattered 4- and 8-byte stores); rows handed out by a ticket counter 110 / 36 with the rows in LDS
// (a workgroup that prefetches holds the ticket of a row it has not started, and higher rows wait for it: one dependency
// chain through all workgroups); one thread per parked entry and the look-back right after the publication 36 -> 14.5;
// one copy of the products in registers instead of two (30 -> 2 spilled registers) 13.1.  Per row then (cycles of thread
// 0, instrumented build): bits 6 k, popcount scan + staging of the next A row 14 k, positions 8 k, row into LDS 7 k,
// product requests 20-24 k, parked products 5 k, look-back 17-20 k, copy-out 5 k: the last two large ones are the memory
// system absorbing every CU's product requests at once (the loaded latency of a state word is ~8 us).
// Tried on top of the 13.1 ms kernel and NOT kept (each bit-identical): the row in two column-range parts with two 512-thread
// workgroups per CU 14.6 (every part repeats the staging, the scans and the look-back with half the threads); the block
// scans on DPP row shifts instead of `__shfl_up` + the binary search with its steps outermost (0 spilled registers) 13.5; a
// second register set so that the next row's products are requested right after its staging instead of after this row is
// assembled 16.5 (the loads of one wave return in order: everything that waits on vmcnt in between - spill reloads, the
// look-back - then waits for the whole burst); without the look-back (wrong offsets, timing only) 12.6: the coupling of the
// workgroups costs 0.6 ms; nothing stored at all 13.0 (the result stores are hidden); no product loads at all (computed
// columns, no parked products) 8.9: the LDS phases are two thirds of the time, the exposed part of the product loads one
// third; waiting for the products before the row's stores are issued (so that no later wait covers the stores) 13.3: no change.
// Model: per row and CU ~15 k cycles of product loads, ~15 k of result stores and ~35 k of LDS
// phases run one after the other (64 k cycles per row); the fabric moves 30 GB per product at 2.4 TB/s.
// Round 5 (13.3 -> 11.2 ms, bit-identical; docs/history/r05.md): the numbers above are round 4's.  Since then a product's A
// element comes from a chunk table (one search per 64 products) instead of a search per product, a position is read from
// 16 bytes (a count per half group) instead of 32, the block scans and the parked products' reductions run on DPP instead of
// ds_bpermute, and no instance spills a register.  Per row now ~28 k cycles of LDS phases + ~25 k of memory time, in sequence.
// Limits (checked by the host before the launch, spamd_spgemm_bitmap_limits): n_col <= 2^20 (8-byte values: ~1.02e6), A rows of at most 256
// elements, at most 1024 * ITEMS products per row (ITEMS = 16 for 4-byte values, 8 for 8-byte ones: the row must fit the
// LDS region), index arrays of either width.  A row whose parked products exceed the list (512 entries) sets the `failed`
// word: the caller then discards the result and uses spgemm_rows.hip.  So does a B operand that is not canonical: a row with
// unsorted columns (split form: a product lands outside its part's range) or with a column twice (equal parked keys).
#include <mutex>

#include "common.h"

namespace spamd {

constexpr int BMK_THREADS = 1024;                           // the wide form (one workgroup per CU); the split form has 512
constexpr int BMK_STAGE = 256;                              // A elements a row may have
constexpr int BMK_MAX_GROUPS = 4096;                        // groups of 256 columns (8 bitmap words)
constexpr int BMK_GPT = 4;                                  // groups per thread = 16-bit fields of the packed scan
constexpr int BMK_CHUNKS = 256;                             // 64-product chunks of a row: 1024 threads x 16 products
constexpr int BMK_LEN_BITS = 23;                            // staging scan: B-row lengths (clamped) below, rows that are not empty above
constexpr int BMK_DUP = 512;                                // parked products per row (wide form; the split form: 256)
constexpr unsigned BMK_NONE = 0xffffffffu;
constexpr int BMK_HEADER = 32;                              // words of `work` before the per-row state words
#ifdef BMK_PROF
#define BMK_T(k) { if (tid == 0) { const unsigned long long now = __builtin_readcyclecounter(); prof[k] += now - tprev; tprev = now; } }
#else
#define BMK_T(k)
#endif

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename V>
struct BmkDup {
  unsigned key;   // (column << 8) | index of the A element
  unsigned rank;  // position of the column in the row (first arrivers), BMK_NONE otherwise
  V val;
};

template <typename V>
struct BmkStage {   // one A row, its elements with an EMPTY B row left out: prefix[e] = products of the elements before e
  unsigned short prefix[BMK_STAGE + 4];   // (strictly ascending, <= 16384; prefix[number of elements] = products of the row)
  int64_t bstart[BMK_STAGE];          // start of B row k_e
  V aval[BMK_STAGE];                  // A value
  unsigned char cst[BMK_CHUNKS];      // element of product 64 * k: chunk k of the row's products is what ONE wave requests at a time
};

struct BmkMisc {
  unsigned long long wa[20];
  unsigned wb[20];
  int ndup[2];    // parked products of the current row ([row parity]: the other one is cleared for the next row meanwhile)
  int64_t row_off;
};

template <typename V>
struct BmkItems {   // products per thread: a row's (column, value) pairs must fit the front region of LDS (see BmkLayout)
  static constexpr int value = sizeof(V) == 4 ? 16 : 8;
};

// LDS: [front region][parked list][filter][two A-row stages][misc].  The front region is the bitmap + the group
// positions while a row's columns are ranked, and the row itself - (column, value) in output order - from then on: it is
// copied out with coalesced stores (scattered 4- and 8-byte stores straight to HBM were measured at 107 ms per product
// at config 5: every one becomes a partial-line write that the L2 cannot combine before it evicts the line).
template <typename V, int THREADS, int ITEMS, int DUP>
struct BmkLayout {
  static constexpr size_t row_bytes = (size_t)THREADS * ITEMS * (4 + sizeof(V));
  __host__ __device__ static size_t bitmap_bytes(int ngroups) { return (size_t)ngroups * 32; }
  __host__ __device__ static size_t pref_bytes(int ngroups) { return ((size_t)ngroups * 4 + 15) / 16 * 16; }   // two per group
  __host__ __device__ static size_t front_bytes(int ngroups) {
    const size_t a = bitmap_bytes(ngroups) + pref_bytes(ngroups);
    return a > row_bytes ? a : row_bytes;
  }
  static size_t bytes(int ngroups) {
    return front_bytes(ngroups) + sizeof(BmkDup<V>) * DUP + DUP * 4 + 2 * sizeof(BmkStage<V>) + sizeof(BmkMisc) + 64;
  }
};

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned bmk_dpp0(unsigned src) {   // lanes without a source read 0
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)src, CTRL, ROW_MASK, 0xf, true);
}
// inclusive scan of three words per lane over the wave (ROWS = 4) or over the first row of 16 lanes (ROWS = 1) on DPP row shifts
template <int ROWS>
__device__ __forceinline__ void bmk_wave_scan3(unsigned& x, unsigned& y, unsigned& z) {
#define BMK_STEP(CTRL, MASK) { const unsigned tx = bmk_dpp0<CTRL, MASK>(x), ty = bmk_dpp0<CTRL, MASK>(y), tz = bmk_dpp0<CTRL, MASK>(z); x += tx; y += ty; z += tz; }
  BMK_STEP(0x111, 0xf)   // row_shr:1
  BMK_STEP(0x112, 0xf)
  BMK_STEP(0x114, 0xf)
  BMK_STEP(0x118, 0xf)
  if constexpr (ROWS == 4) {
    BMK_STEP(0x142, 0xa)   // row_bcast15 -> rows 1, 3
    BMK_STEP(0x143, 0xc)   // row_bcast31 -> rows 2, 3
  }
#undef BMK_STEP
}
// exclusive block scan of (a: four packed 16-bit counts - no field overflows, so the two halves add independently -, b: one
// 32-bit count); totals in ta / tb.  Two LDS barriers.  (Round 4 went through `__shfl_up`, a ds_bpermute per step and word:
// 6.5 k of a row's cycles; on DPP 1.7 k.)
template <int THREADS>
__device__ __forceinline__ void bmk_block_scan(unsigned long long& a, unsigned& b, unsigned long long& ta, unsigned& tb,
                                               BmkMisc* m) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  constexpr int NW = THREADS / 64;
  static_assert(NW <= 16, "the wave totals are scanned inside one row of 16 lanes");
  unsigned x = (unsigned)a, y = (unsigned)(a >> 32), z = b;
  bmk_wave_scan3<4>(x, y, z);
  if (lane == 63) {
    m->wa[wid] = ((unsigned long long)y << 32) | x;
    m->wb[wid] = z;
  }
  lds_barrier();
  if (wid == 0) {
    const unsigned long long wa = lane < NW ? m->wa[lane] : 0;
    const unsigned wb = lane < NW ? m->wb[lane] : 0;
    unsigned sx = (unsigned)wa, sy = (unsigned)(wa >> 32), sz = wb;
    bmk_wave_scan3<1>(sx, sy, sz);
    const unsigned long long sa = ((unsigned long long)sy << 32) | sx;
    if (lane < NW) {
      m->wa[lane] = sa - wa;     // (field-wise: no borrow crosses a field, the inclusive sums are >= their terms)
      m->wb[lane] = sz - wb;
    }
    if (lane == NW - 1) {
      m->wa[NW] = sa;
      m->wb[NW] = sz;
    }
  }
  lds_barrier();
  const unsigned long long incl = ((unsigned long long)y << 32) | x;
  a = incl - a + m->wa[wid];
  b = z - b + m->wb[wid];
  ta = m->wa[THREADS / 64];
  tb = m->wb[THREADS / 64];
}

// set bits of group g (8 words = 256 columns): the first four words' count in `lo`, all eight returned
__device__ __forceinline__ int bmk_popc_group(const unsigned* bm, int g, int& lo_cnt) {
  const uint4 lo = *reinterpret_cast<const uint4*>(bm + (size_t)g * 8);
  const uint4 hi = *reinterpret_cast<const uint4*>(bm + (size_t)g * 8 + 4);
  lo_cnt = __popc(lo.x) + __popc(lo.y) + __popc(lo.z) + __popc(lo.w);
  return lo_cnt + __popc(hi.x) + __popc(hi.y) + __popc(hi.z) + __popc(hi.w);
}

// number of set bits below column `col`: the column's position in the sorted row.  pref[] holds the bits below every HALF
// group (128 columns = one 16-byte read; round 4 kept one per 256 columns and read 32 bytes per product: the positions were
// the phase with the most LDS traffic)
__device__ __forceinline__ int bmk_rank(const unsigned* bm, const unsigned short* pref, unsigned col) {
  const unsigned h = col >> 7, wi = (col >> 5) & 3u;
  const uint4 q = *reinterpret_cast<const uint4*>(bm + (size_t)h * 4);
  const unsigned w[4] = {q.x, q.y, q.z, q.w};
  int r = (int)pref[h];
  unsigned cur = w[0];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r += (unsigned)i < wi ? __popc(w[i]) : 0;
    cur = (unsigned)i == wi ? w[i] : cur;
  }
  return r + __popc(cur & ((1u << (col & 31u)) - 1u));
}

// common.h's decoupled look-back with the aggregate already published by the caller, FOUR windows of 64 predecessors per
// round trip (a state word written by another XCD comes from the fabric: ~2 us each; with one workgroup per CU the
// nearest row with a known prefix is up to 255 rows back, i.e. four dependent round trips with one window at a time:
// 17 k of a row's 91 k cycles), and a back-off in the spin.
// BOUNDED: the rows this one waits for are held by workgroups that must be resident (the grid is one workgroup per CU - or
// the occupancy query's number - so they are, unless CUs are masked away from the process or held by another stream's
// kernels for seconds); after ~2^21 polls (seconds) the wait is given up, `gave_up` is set and the caller fails the call,
// whose result the host then discards.
__device__ __forceinline__ unsigned long long bmk_lookback(unsigned long long* st, int64_t blk, unsigned long long tot, int lane,
                                                           bool& gave_up) {
  const unsigned long long mask = (1ull << 62) - 1;
  unsigned long long excl = 0;
  int64_t hi = blk - 1;
  int polls = 0;
  while (hi >= 0) {
    unsigned long long v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int64_t j = hi - lane - 64 * t;
      v[t] = 2ull << 62;   // rows before row 0 behave like "prefix known, value 0"
      if (j >= 0) v[t] = __hip_atomic_load(&st[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    bool done = false, retry = false;
    unsigned long long part = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (!done && !retry) {   // (wave-uniform)
        const unsigned long long flag = v[t] >> 62;
        const unsigned long long have_prefix = __ballot(flag == 2);
        const unsigned long long missing = __ballot(flag == 0);
        const int first_prefix = have_prefix ? __builtin_ctzll(have_prefix) : 64;
        const unsigned long long upto = first_prefix >= 63 ? ~0ull : ((2ull << first_prefix) - 1);
        if (missing & upto) {
          retry = true;
        } else {
          part += lane <= first_prefix ? (v[t] & mask) : 0;
          done = first_prefix < 64;
        }
      }
    }
    if (retry) {   // (what was summed so far is dropped: the same windows are read again)
      if (++polls > (1 << 21)) {
        gave_up = true;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
      continue;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
    excl += part;
    if (done) break;
    hi -= 256;
  }
  if (lane == 0)
    __hip_atomic_store(&st[blk], (2ull << 62) | (excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return excl;
}

template <typename V>
__device__ __forceinline__ int bmk_is_zero_bits(V v) {
  if constexpr (sizeof(V) == 8) return __builtin_bit_cast(unsigned long long, v) == 0;
  else return __builtin_bit_cast(unsigned, v) == 0;
}

// One workgroup works on one PART of an output row at a time: the columns [h * range, (h + 1) * range) of row r, h < np
// ("virtual row" v = r * np + h; np = 1: whole rows).  Parts of one row are emitted one behind the other, so the look-back
// runs over the virtual rows and the result is the same CSR.  With np > 1, `bsplit[k * (np - 1) + h]` = the first element of
// B row k whose column is >= (h + 1) * range (spgemm_bsplit_kernel): a part's products are contiguous pieces of B rows.
template <typename V, typename I, int ITEMS, int THREADS, int DUP, bool SPLIT>
__global__ void __launch_bounds__(THREADS, 4)   // (four waves per SIMD: one 1024-thread or two 512-thread workgroups per CU)
spgemm_bitmap_kernel(int64_t n_vrow, int np_arg, int64_t range, int ngroups, const I* __restrict__ a_ptr, const I* __restrict__ a_idx,
                     const V* __restrict__ a_val, const I* __restrict__ b_ptr, const I* __restrict__ bsplit,
                     const I* __restrict__ b_idx, const V* __restrict__ b_val, unsigned long long* __restrict__ work,
                     int64_t* __restrict__ out_ptr, int64_t* __restrict__ out_idx, V* __restrict__ out_val) {
#pragma clang fp contract(off)
  using L = BmkLayout<V, THREADS, ITEMS, DUP>;
  const int np = SPLIT ? np_arg : 1;                    // (whole rows: every `np > 1` branch below folds away)
  constexpr int FILT_WORDS = DUP;                       // 32 filter bits per list entry
  constexpr unsigned FILT_MASK = FILT_WORDS * 32 - 1;
  static_assert(BMK_GPT * THREADS * 256 >= (1 << 19), "column range of a part");
  static_assert(ITEMS % 4 == 0, "A-element indices are packed four to a register");
  static_assert(THREADS * ITEMS <= 65535, "the staged prefix is 16 bits wide");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned* const bm = reinterpret_cast<unsigned*>(smem);
  unsigned short* const pref = reinterpret_cast<unsigned short*>(smem + L::bitmap_bytes(ngroups));
  BmkDup<V>* const dup = reinterpret_cast<BmkDup<V>*>(smem + L::front_bytes(ngroups));
  // the finished row in output order: 4-byte values as {column, value bits} pairs, 8-byte values as two arrays
  uint2* const row_cv = reinterpret_cast<uint2*>(smem);
  unsigned* const row_c = reinterpret_cast<unsigned*>(smem);
  V* const row_v = reinterpret_cast<V*>(smem + (size_t)THREADS * ITEMS * 4);
  auto put = [&](unsigned at, unsigned c, V v) {
    if constexpr (sizeof(V) == 4) {
      row_cv[at] = make_uint2(c, __builtin_bit_cast(unsigned, v));
    } else {
      row_c[at] = c;
      row_v[at] = v;
    }
  };
  unsigned* const filt = reinterpret_cast<unsigned*>(dup + DUP);
  BmkStage<V>* const stage = reinterpret_cast<BmkStage<V>*>(filt + FILT_WORDS);
  BmkMisc* const misc = reinterpret_cast<BmkMisc*>(stage + 2);
  unsigned long long* const state = work + BMK_HEADER;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wid_s = __builtin_amdgcn_readfirstlane(wid);
  constexpr int CAP = THREADS * ITEMS;

  // ---- set-up: clean LDS, the first two tickets -------------------------------------------------------------------------
  for (int i = tid; i < ngroups * 2; i += THREADS) reinterpret_cast<uint4*>(bm)[i] = make_uint4(0, 0, 0, 0);
  for (int i = tid; i < FILT_WORDS; i += THREADS) filt[i] = 0;
  if (tid == 0) misc->ndup[0] = misc->ndup[1] = 0;
  lds_barrier();
  // Rows are dealt round-robin: workgroup w takes rows w, w + G, w + 2 G, ... (G = the grid = one workgroup per CU, all
  // resident).  A row's look-back then only ever waits for rows that the OTHER workgroups are working on at the same time,
  // or that are finished.  (Tickets taken from a counter were built first and measured 4x SLOWER than the bucket kernels:
  // with a row prefetched, a workgroup holds the ticket of a row it has not started while higher rows, held by others,
  // already wait for it in their look-back - the rows of all workgroups end up in one dependency chain.)
  const int64_t G = gridDim.x;
  int64_t cur = blockIdx.x, nxt = cur + G;
  bool failed = false;

  // the A row of `row`: element `tid` (column of A = row of B, value); rows longer than the staging area fail the call
  auto load_arow = [&](int64_t vrow, int& nA, int64_t& ka, V& av) {
    nA = 0;
    ka = 0;
    av = V(0);
    if (vrow < n_vrow) {
      const int64_t row = np > 1 ? vrow / np : vrow;
      const int64_t a0 = (int64_t)a_ptr[row];
      const int64_t n = (int64_t)a_ptr[row + 1] - a0;
      if (n > BMK_STAGE) failed = true;
      else nA = (int)n;
      if (tid < nA) {
        ka = (int64_t)a_idx[a0 + tid];
        av = a_val[a0 + tid];
      }
    }
  };
  auto load_brow = [&](int64_t vrow, int nA, int64_t ka, int64_t& bs, unsigned& len) {
    bs = 0;
    len = 0;
    if (tid < nA) {
      if (np > 1) {   // the piece of B row ka inside this part's column range
        const int h = (int)(vrow % np);
        const I* const sp = bsplit + ka * (np - 1);
        bs = h == 0 ? (int64_t)b_ptr[ka] : (int64_t)sp[h - 1];
        len = (unsigned)((h == np - 1 ? (int64_t)b_ptr[ka + 1] : (int64_t)sp[h]) - bs);
      } else {
        bs = (int64_t)b_ptr[ka];
        len = (unsigned)((int64_t)b_ptr[ka + 1] - bs);
      }
    }
  };
  // products of the staged row -> registers (loads only: nothing here waits for them).  Product p belongs to thread
  // p % 1024: for one item index consecutive lanes read consecutive entries of a B row.
  unsigned colN[ITEMS];
  V bvN[ITEMS];
  unsigned eN[ITEMS / 4];
  // chunk table of a staged row (its prefix must be visible): the last element e with prefix[e] <= 64 * k, one branch-free
  // binary search per chunk by the last waves of the workgroup (wave 0 has the look-back)
  auto build_cst = [&](BmkStage<V>* st, int nE) {
    constexpr int NCH = CAP / 64;
    static_assert(NCH <= BMK_CHUNKS && NCH <= THREADS, "chunk table");
    const int k = tid - (THREADS - NCH);
    if (k >= 0) {
      const int p = k * 64;
      int e = 0;
#pragma unroll
      for (int step = BMK_STAGE / 2; step >= 1; step >>= 1) {
        const int t = e + step;              // (<= 255: always inside the staging array, whatever nE is)
        const int at_t = (int)st->prefix[t];
        e = ((t < nE) & (at_t <= p)) ? t : e;
      }
      st->cst[k] = (unsigned char)e;
    }
  };
  // Product p = 64 * k + lane of the row belongs to lane `lane` of the wave that takes chunk k: item j of wave w is chunk
  // j * (THREADS / 64) + w (= product j * THREADS + tid: for one item index consecutive lanes read consecutive entries of a
  // B row).  The element of the chunk's first product comes from the table; the prefix is strictly ascending, so at most 63
  // element boundaries lie inside the chunk, and they are the first of the 64 that follow that element: lane l fetches
  // boundary l, and a product's element is the table's plus the boundaries at or below it - four of them straight-line
  // (rows of B with ~100 elements: 0-2 per chunk), the rest in a wave-uniform loop.  (Round 4 searched the prefix per
  // product: 8 dependent LDS reads x 16 items per thread, a quarter of the kernel's LDS instructions.)
  auto expand = [&](const BmkStage<V>* st, int nE, int P) {
    // (an opaque zero: without it the sixteen product numbers and chunk ends are loop invariants of the row loop, the
    // compiler keeps them in registers across it and spills - and a spill reload among the product requests waits for all of
    // them: vmcnt counts in order)
    int z;
    asm volatile("s_mov_b32 %0, 0" : "=s"(z));
    const int pt = tid + z;                 // product of item 0
    const int cw = wid_s * 64 + z;          // its chunk's first product (wave-uniform)
    const unsigned char* const cstp = st->cst + wid;
#pragma unroll
    for (int j = 0; j < ITEMS / 4; ++j) eN[j] = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      const int p = pt + j * THREADS;
      colN[j] = BMK_NONE;
      bvN[j] = V(0);
      const int e0 = cstp[j * (THREADS / 64)];
      const int bi = e0 + 1 + lane;
      const int b = bi <= nE ? (int)st->prefix[bi] : 0x7fffffff;   // (a boundary beyond the chunk is above every p of it)
      int e = e0 + (p >= __builtin_amdgcn_readlane(b, 0)) + (p >= __builtin_amdgcn_readlane(b, 1)) +
              (p >= __builtin_amdgcn_readlane(b, 2)) + (p >= __builtin_amdgcn_readlane(b, 3));
      unsigned long long more = __ballot(b <= cw + j * THREADS + 63) >> 4;
      for (int i = 4; more & 1ull; more >>= 1, ++i) e += p >= __builtin_amdgcn_readlane(b, i);
      if (p < P) {
        const int64_t q = st->bstart[e] + (int64_t)(p - (int)st->prefix[e]);
        colN[j] = (unsigned)b_idx[q];
        bvN[j] = b_val[q];
        eN[j / 4] |= (unsigned)e << (8 * (j % 4));
      }
      if (j % 8 == 7) __builtin_amdgcn_sched_barrier(0);   // (keeps the scheduler from hoisting all 2 x ITEMS address chains)
    }
  };
  // the staging scan sums (B-row length clamped to CAP + 1: 256 of them stay below 2^23) | (1 << 23 for a row that is not empty)
  auto scan_word = [&](unsigned len) -> unsigned {
    return (len > (unsigned)CAP ? (unsigned)CAP + 1u : len) | (len ? 1u << BMK_LEN_BITS : 0u);
  };
  // one thread per A element puts it at its place among the elements whose B row is not empty
  auto stage_row = [&](BmkStage<V>* st, unsigned excl, unsigned tot, unsigned len, int64_t bs, V av, int& nE, int& P) {
    unsigned tl = tot & ((1u << BMK_LEN_BITS) - 1u);
    nE = (int)(tot >> BMK_LEN_BITS);
    if (tl > (unsigned)CAP) {
      failed = true;
      tl = 0;
    }
    P = (int)tl;
    if (len) {
      const int ce = (int)(excl >> BMK_LEN_BITS);
      st->prefix[ce] = tl ? (unsigned short)(excl & ((1u << BMK_LEN_BITS) - 1u)) : (unsigned short)0;
      st->bstart[ce] = bs;
      st->aval[ce] = av;
    }
    if (tid == 0) st->prefix[nE] = (unsigned short)tl;
  };

  // ---- prologue: stage the first row, request its products, fetch the second row's A elements ---------------------------
  int nA_n, nE_c, P_c;   // A elements of the next row (to fetch), staged elements and products of the current one
  int64_t ka;
  V av;
  {
    int64_t bs;
    unsigned len;
    int nA_c;
    load_arow(cur, nA_c, ka, av);
    load_brow(cur, nA_c, ka, bs, len);
    unsigned long long za = 0, ta;
    unsigned excl = scan_word(len), tl;
    bmk_block_scan<THREADS>(za, excl, ta, tl, misc);
    stage_row(&stage[0], excl, tl, len, bs, av, nE_c, P_c);
    lds_barrier();
    build_cst(&stage[0], nE_c);
    lds_barrier();
    expand(&stage[0], nE_c, P_c);
    load_arow(nxt, nA_n, ka, av);
  }
  int buf = 0;
  int zero_count = 0;
#ifdef BMK_PROF
  unsigned long long prof[16] = {0};
  unsigned long long tprev = __builtin_readcyclecounter();
#endif

  while (cur < n_vrow) {   // (workgroup-uniform)
    const BmkStage<V>* const sc = &stage[buf];
    BmkStage<V>* const sn = &stage[buf ^ 1];
    int* const ndup = &misc->ndup[buf];
    // ---- top: the B row pointers of the next row's A elements ------------------------------------------------------------
    const int64_t nn = nxt + G;
    const unsigned cbase = np > 1 ? (unsigned)((cur % np) * range) : 0u;   // first column of this part
    BMK_T(0)
    int64_t bs;
    unsigned len;
    load_brow(nxt, nA_n, ka, bs, len);
    const V av_n = av;
    // ---- 1. every product of the current row sets its column's bit -----------------------------------------------------
    // (the products stay where the prefetch put them - colN / bvN / eN - until the row is assembled in step 4: a second
    // copy of them, as keys and values, had the kernel spill 30 registers)
    unsigned first_mask = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      // (a column outside this part's range: B's row is not sorted by column - the split form's binary search over it put the
      // product into the wrong part.  Reachable through `GCXS((data, indices, indptr))`, which takes the caller's arrays as
      // they are: the product is dropped BEFORE it touches the bitmap and the call fails; the host then takes the bucket
      // kernels, which do not depend on B's order.  The wide form's columns are always inside [0, n_col).)
      if (SPLIT && colN[j] != BMK_NONE && colN[j] - cbase >= (unsigned)range) {
        failed = true;
        colN[j] = BMK_NONE;
      }
      if (colN[j] != BMK_NONE) {
        const unsigned c = colN[j] - cbase;
        const unsigned bit = 1u << (c & 31u);
        const unsigned old = atomicOr(&bm[c >> 5], bit);
        if (old & bit) {   // the output element has a product already: park this one
          const unsigned e = (eN[j / 4] >> (8 * (j % 4))) & 255u;
          const unsigned h = (c ^ (c >> 14)) & FILT_MASK;
          atomicOr(&filt[h >> 5], 1u << (h & 31u));
          const int slot = atomicAdd(ndup, 1);
          if (slot < DUP) {
            dup[slot].key = (c << 8) | e;
            dup[slot].rank = BMK_NONE;
            dup[slot].val = sc->aval[e] * bvN[j];
          }
        } else {
          first_mask |= 1u << j;
        }
      }
      if (j % 8 == 7) __builtin_amdgcn_sched_barrier(0);
    }
    BMK_T(1)
    lds_barrier();
    BMK_T(2)
    // ---- 2. popcount scan: positions of the columns; the same scan sums the next row's B-row lengths --------------------
    unsigned long long cnts = 0;
    unsigned los = 0;   // bits of the first half of each of the thread's groups (<= 128: a byte each)
#pragma unroll
    for (int m = 0; m < BMK_GPT; ++m) {
      const int g = tid + THREADS * m;
      if (g < ngroups) {
        int lo_cnt;
        cnts |= (unsigned long long)bmk_popc_group(bm, g, lo_cnt) << (16 * m);
        los |= (unsigned)lo_cnt << (8 * m);
      }
    }
    unsigned long long excl_c = cnts, tot_c;
    unsigned excl_l = scan_word(len), tot_l;
    bmk_block_scan<THREADS>(excl_c, excl_l, tot_c, tot_l, misc);
    int row_nnz = 0;
#pragma unroll
    for (int m = 0; m < BMK_GPT; ++m) {
      const int g = tid + THREADS * m;
      if (g < ngroups) {
        const unsigned below = (unsigned)row_nnz + (unsigned)((excl_c >> (16 * m)) & 0xffffu);
        reinterpret_cast<unsigned*>(pref)[g] = below | ((below + ((los >> (8 * m)) & 0xffu)) << 16);
      }
      row_nnz += (int)((tot_c >> (16 * m)) & 0xffffu);
    }
    int nE_n, P_n;
    stage_row(sn, excl_l, tot_l, len, bs, av_n, nE_n, P_n);
    BMK_T(3)
    lds_barrier();
    BMK_T(4)
    build_cst(sn, nE_n);   // (visible to the expansion in step 4: one more barrier on the way)
    // ---- 3. the row's length is known: wave 0 publishes it and looks back for the row's offset, while every first
    // arriver finds its column's position (unless products of its column are parked: then it joins them) ---------------
    // (only the publication happens here: the look-back itself waits until the row is assembled - step 5 -, by which time
    // the rows before this one have published theirs; looking back right here cost 61 k of a row's 194 k cycles)
    if (tid == 0 && cur > 0)
      __hip_atomic_store(&state[cur], (1ull << 62) | (unsigned long long)row_nnz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    BMK_T(5)
    // positions: 14 bits each, two to a register
    unsigned rk[ITEMS / 2];
#pragma unroll
    for (int j = 0; j < ITEMS / 2; ++j) rk[j] = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      if ((first_mask >> j) & 1u) {
        const unsigned c = colN[j] - cbase;
        const unsigned h = (c ^ (c >> 14)) & FILT_MASK;
        const bool parked = (filt[h >> 5] >> (h & 31u)) & 1u;
        const int r = bmk_rank(bm, pref, c);
        rk[j / 2] |= (unsigned)r << (16 * (j % 2));
        if (parked) {
          first_mask &= ~(1u << j);
          const unsigned e = (eN[j / 4] >> (8 * (j % 4))) & 255u;
          const int slot = atomicAdd(ndup, 1);
          if (slot < DUP) {
            dup[slot].key = (c << 8) | e;
            dup[slot].rank = (unsigned)r;
            dup[slot].val = sc->aval[e] * bvN[j];
          }
        }
      }
      if (j % 4 == 3) __builtin_amdgcn_sched_barrier(0);   // (four lookups = 36 words in flight, not 16 x 9)
    }
    BMK_T(6)
    lds_barrier();
    BMK_T(7)
    // ---- 4. every position is known and the bitmap has served: the front region now takes the row, (column, value) at its
    // position; then the next row's products are requested (they land while this row is finished and written) -----------
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      if ((first_mask >> j) & 1u) {
        const unsigned e = (eN[j / 4] >> (8 * (j % 4))) & 255u;
        put((rk[j / 2] >> (16 * (j % 2))) & 0xffffu, colN[j], sc->aval[e] * bvN[j]);
      }
    }
    BMK_T(8)
    expand(sn, nE_n, P_n);
    BMK_T(9)
    // ---- 5. parked products: a WAVE per entry scans the list (8 entries per lane at most); the entry with the smallest
    // A-element index of its column adds the column's products left to right, in the order of A's elements ------------
    {
      int n = *ndup;
      if (n > DUP) {
        failed = true;
        n = DUP;
      }
      constexpr int PER = DUP / 64;
      unsigned lk[PER];
#pragma unroll
      for (int t = 0; t < PER; ++t) lk[t] = lane + 64 * t < n ? dup[lane + 64 * t].key : BMK_NONE;
      for (int d = wid; d < n; d += THREADS / 64) {
        const unsigned kd = dup[d].key;          // (wave-uniform)
        const unsigned c = kd >> 8;
        // smallest key of the column, and where the column's position is recorded (exactly one entry has it)
        unsigned lo = BMK_NONE;
        int cnt = 0, holder = -1;
#pragma unroll
        for (int t = 0; t < PER; ++t) {
          const bool m = lk[t] != BMK_NONE && (lk[t] >> 8) == c;
          lo = m && lk[t] < lo ? lk[t] : lo;
          cnt += m;
        }
        lo = wave_min_u32(lo);                   // (DPP: round 4 reduced through ds_bpermute - a parked entry cost ~2 k cycles)
        cnt = (int)wave_sum_u32((unsigned)cnt);
        if (lo != kd) continue;                  // not the first A element of this column (wave-uniform)
        V acc = dup[d].val;
        unsigned rank = dup[d].rank;
        unsigned last = kd;
        for (int step = 1; step < cnt; ++step) {   // the next larger A-element index of this column, cnt - 1 times
          unsigned nx = BMK_NONE;
          int at = 0;
#pragma unroll
          for (int t = 0; t < PER; ++t) {
            const bool m = lk[t] != BMK_NONE && (lk[t] >> 8) == c && lk[t] > last && lk[t] < nx;
            nx = m ? lk[t] : nx;
            at = m ? lane + 64 * t : at;
          }
          const unsigned best = wave_min_u32(nx);
          const unsigned long long who = __ballot(nx == best && best != BMK_NONE);
          if (who == 0) {   // fewer distinct keys than entries of this column: a B row holds the column TWICE (equal
            failed = true;  // (column, A element) keys) - not a canonical operand; fail the call, the bucket kernels take it
            break;
          }
          const int src = __builtin_amdgcn_readlane(at, (int)__builtin_ctzll(who));
          acc = acc + dup[src].val;
          const unsigned r2 = dup[src].rank;
          rank = r2 != BMK_NONE ? r2 : rank;
          last = best;
        }
        if (lane == 0 && rank != BMK_NONE) put(rank, c + cbase, acc);   // (always a position, unless the list overflowed: failed anyway)
      }
    }
    BMK_T(10)
    // the row's offset in the result: sum of the lengths of the rows before it (decoupled look-back, wave 0).  Measured
    // in three places - right after the publication, before the product requests, here - it takes ~17 k cycles wherever it
    // stands: the state words travel through a memory system that every CU has just filled with its product requests.
    if (wid == 0) {
#if defined(BMK_ABL) && BMK_ABL == 1   // timing ablation (wrong row offsets): no look-back, rows at a fixed pitch
      const unsigned long long before = (unsigned long long)cur * 9900ull;
#else
      bool gave_up = false;
      const unsigned long long before = bmk_lookback(state, cur, (unsigned long long)row_nnz, lane, gave_up);
      if (gave_up) failed = true;
#endif
      if (lane == 0) {
        misc->row_off = (int64_t)before;
        if (np == 1) out_ptr[cur + 1] = (int64_t)before + row_nnz;
        else if (cur % np == np - 1) out_ptr[cur / np + 1] = (int64_t)before + row_nnz;
        if (cur == 0) out_ptr[0] = 0;
      }
    }
    BMK_T(14)
    int nA_nn;
    load_arow(nn, nA_nn, ka, av);
    // ---- 6. the row leaves with coalesced stores; the bitmap's bytes are left zeroed for the next row ---------------------
    lds_barrier();
    BMK_T(11)
    const int64_t row_off = misc->row_off;
    {
      const int units = ngroups * 4;   // the bitmap in 8-byte units
      if constexpr (sizeof(V) == 4) {
        // a thread clears exactly the 8 bytes it has just read: no barrier between copying out and clearing
        for (int u = tid; u < (units > row_nnz ? units : row_nnz); u += THREADS) {
          if (u < row_nnz) {
            const uint2 cv = row_cv[u];
            out_idx[row_off + u] = (int64_t)cv.x;
            out_val[row_off + u] = __builtin_bit_cast(V, cv.y);
            zero_count += cv.y == 0;
          }
          if (u < units) row_cv[u] = make_uint2(0, 0);
        }
      } else {
        for (int u = tid; u < row_nnz; u += THREADS) {
          const V v = row_v[u];
          out_idx[row_off + u] = (int64_t)row_c[u];
          out_val[row_off + u] = v;
          zero_count += bmk_is_zero_bits(v);
        }
        lds_barrier();
        for (int u = tid; u < units; u += THREADS) row_cv[u] = make_uint2(0, 0);
      }
    }
    for (int i = tid; i < FILT_WORDS; i += THREADS) filt[i] = 0;
    if (tid == 0) misc->ndup[buf ^ 1] = 0;   // (the previous row's count: read for the last time two barriers ago)
    BMK_T(12)
    lds_barrier();
    BMK_T(13)
    cur = nxt;
    nxt = nn;
    nA_n = nA_nn;
    nE_c = nE_n;
    P_c = P_n;
    buf ^= 1;
  }
#ifdef BMK_PROF
  if (tid == 0)
    for (int k = 0; k < 16; ++k) atomicAdd(work + 4 + k, prof[k]);
#endif
  // ---- epilogue: exact zeros written, failure word ------------------------------------------------------------------------
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) zero_count += __shfl_xor(zero_count, d, 64);
  if (lane == 0 && zero_count) atomicAdd(work + 2, (unsigned long long)zero_count);
  // (any lane: a product outside its part's column range is seen by the one thread that holds it)
  if (__ballot(failed) != 0 && lane == 0) __hip_atomic_store(work + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// bsplit[k * (np - 1) + h] = first element of B row k whose column is >= (h + 1) * range (h < np - 1)
template <typename I>
__global__ void __launch_bounds__(256) spgemm_bsplit_kernel(int64_t n_inner, int np, int64_t range, const I* __restrict__ b_ptr,
                                                            const I* __restrict__ b_idx, I* __restrict__ bsplit) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_inner * (np - 1)) return;
  const int64_t k = t / (np - 1);
  const int64_t target = (t % (np - 1) + 1) * range;
  int64_t lo = (int64_t)b_ptr[k], hi = (int64_t)b_ptr[k + 1];
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)b_idx[mid] < target) lo = mid + 1;
    else hi = mid;
  }
  bsplit[t] = (I)lo;
}

// the two forms: WIDE = whole rows, one 1024-thread workgroup per CU, n_col <= 2^20; SPLIT = parts of rows (column ranges),
// 512 threads and <= 80 KB of LDS, so that TWO workgroups share a CU and fill each other's barrier and memory waits
constexpr int BMK_SPLIT_THREADS = 512, BMK_SPLIT_ITEMS = 16, BMK_SPLIT_DUP = 256;   // (8-byte values: 8 products per thread)
template <typename V>
struct BmkSplitItems {
  static constexpr int value = sizeof(V) == 4 ? BMK_SPLIT_ITEMS : BMK_SPLIT_ITEMS / 2;
};

template <typename V>
static int64_t bmk_wide_max_groups() {   // groups of 256 columns whose bitmap + positions fit next to the rest in 160 KB
  using L = BmkLayout<V, BMK_THREADS, BmkItems<V>::value, BMK_DUP>;
  int64_t g = BMK_MAX_GROUPS;
  while (g > 0 && (int64_t)L::bytes((int)g) > 160 * 1024) --g;
  return g;
}

template <typename V>
static int64_t bmk_split_max_groups() {   // groups of 256 columns whose bitmap + positions fit next to the rest in 80 KB
  using L = BmkLayout<V, BMK_SPLIT_THREADS, BmkSplitItems<V>::value, BMK_SPLIT_DUP>;
  const int64_t rest = (int64_t)L::bytes(0) - (int64_t)L::front_bytes(0);
  int64_t g = (80 * 1024 - rest) / 36;
  while (g > 0 && (int64_t)L::bytes((int)g) > 80 * 1024) --g;
  return g;
}

template <typename V, typename I, int ITEMS, int THREADS, int DUP, bool SPLIT>
static int bmk_launch(int64_t n_row, int np, int64_t range, const I* a_ptr, const I* a_idx, const V* a_val, const I* b_ptr,
                      const I* bsplit, const I* b_idx, const V* b_val, unsigned long long* work, int64_t* out_ptr,
                      int64_t* out_idx, V* out_val, hipStream_t s) {
  using L = BmkLayout<V, THREADS, ITEMS, DUP>;
  auto kern = &spgemm_bitmap_kernel<V, I, ITEMS, THREADS, DUP, SPLIT>;
  const int ngroups = (int)ceil_div(range, (int64_t)256);
  if (ngroups > BMK_GPT * THREADS) return SPAMD_EINVAL;
  const size_t lds = L::bytes(ngroups);
  if (lds > 160 * 1024) return SPAMD_EINVAL;
  if (int rc = set_max_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024)) return rc;
  int dev = 0, cus = 0, per_cu = 0;
  if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return (int)e;
  if (hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); e != hipSuccess) return (int)e;
  if (hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, THREADS, lds); e != hipSuccess) return (int)e;
  // every workgroup of the grid must be resident (a row's look-back waits for the rows the OTHER workgroups hold)
  per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);
  const int64_t n_vrow = n_row * np;
  const int64_t want = (int64_t)cus * per_cu;
  const int64_t grid = n_vrow < want ? n_vrow : want;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(THREADS), lds, s, n_vrow, np, range, ngroups, a_ptr, a_idx, a_val, b_ptr,
                     bsplit, b_idx, b_val, work, out_ptr, out_idx, out_val);
  return launch_status();
}

}  // namespace spamd

using namespace spamd;

// limits: which = 0: products per row (wide form), 1: A elements per row, 2: columns (wide form), 3: parked products per
// row (wide form), 4: products per PART of a row (split form), 5: columns per part (split form), 6: parked products per
// part (split form)
extern "C" int64_t spamd_spgemm_bitmap_limits(int val_dtype, int which) {
  const bool v4 = val_dtype == SPAMD_F32 || val_dtype == SPAMD_I32;
  switch (which) {
    case 0: return (int64_t)BMK_THREADS * (v4 ? BmkItems<float>::value : BmkItems<double>::value);
    case 1: return BMK_STAGE;
    case 2: return (v4 ? bmk_wide_max_groups<float>() : bmk_wide_max_groups<double>()) * 256;
    case 3: return BMK_DUP;
    case 4: return (int64_t)BMK_SPLIT_THREADS * (v4 ? BmkSplitItems<float>::value : BmkSplitItems<double>::value);
    case 5: return (v4 ? bmk_split_max_groups<float>() : bmk_split_max_groups<double>()) * 256;
    case 6: return BMK_SPLIT_DUP;
    default: return -1;
  }
}

// C = A @ B, rows written in place: out_indptr[n_row + 1], out_indices / out_data with room for every product (the
// caller trims to out_indptr[n_row]).  parts = 1: the wide form (n_col <= limit 2).  parts > 1: every row
// in `parts` column ranges of ceil(n_col / parts) columns rounded up to 256 (<= limit 5), two workgroups per CU; `bsplit`
// = n_inner * (parts - 1) words of the index type, filled here (n_inner = rows of B).  work: n_row * parts + 32 words,
// zeroed here; afterwards work[1] != 0 = failed (a row or part outside the limits, or with more parked products than the
// list holds: discard the result), work[2] = values written whose bits are all zero.
extern "C" int spamd_spgemm_bitmap(int val_dtype, int idx_dtype, int64_t n_row, int64_t n_inner, int64_t n_col, int parts,
                                   const void* a_indptr, const void* a_indices, const void* a_data, const void* b_indptr,
                                   const void* b_indices, const void* b_data, void* bsplit, int64_t* work,
                                   int64_t* out_indptr, int64_t* out_indices, void* out_data, void* stream) {
  if (n_row < 0 || n_inner < 0 || n_col <= 0 || parts < 1 || parts > 4096 || !work || !out_indptr) return SPAMD_EINVAL;
  if (parts > 1 && !bsplit) return SPAMD_EINVAL;
  const bool v4 = val_dtype == SPAMD_F32 || val_dtype == SPAMD_I32;
  if (parts == 1 && n_col > (v4 ? bmk_wide_max_groups<float>() : bmk_wide_max_groups<double>()) * 256) return SPAMD_EINVAL;
  int64_t range = ceil_div(ceil_div(n_col, (int64_t)parts), (int64_t)256) * 256;
  if (parts > 1 && range > (v4 ? bmk_split_max_groups<float>() : bmk_split_max_groups<double>()) * 256) return SPAMD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (hipError_t e = hipMemsetAsync(work, 0, (size_t)(n_row * parts + BMK_HEADER) * sizeof(int64_t), s); e != hipSuccess) return (int)e;
  if (n_row == 0) return (int)hipMemsetAsync(out_indptr, 0, sizeof(int64_t), s);
  if (parts > 1 && n_inner > 0) {
    const int64_t n = n_inner * (parts - 1);
    SPAMD_DISPATCH_IDX(idx_dtype, I, hipLaunchKernelGGL(spgemm_bsplit_kernel<I>, dim3((unsigned)ceil_div(n, (int64_t)256)), dim3(256), 0,
                                                        s, n_inner, parts, range, (const I*)b_indptr, (const I*)b_indices, (I*)bsplit))
    if (int rc = launch_status()) return rc;
  }
  unsigned long long* const w = reinterpret_cast<unsigned long long*>(work);
#define BMK_GO(V, I, ITEMS, THREADS, DUP, SPLIT)                                                                                  \
  return (bmk_launch<V, I, ITEMS, THREADS, DUP, SPLIT>(n_row, parts, range, (const I*)a_indptr, (const I*)a_indices, (const V*)a_data, \
                                                (const I*)b_indptr, (const I*)bsplit, (const I*)b_indices, (const V*)b_data, w,  \
                                                out_indptr, out_indices, (V*)out_data, s));
  if (parts > 1) {
    SPAMD_DISPATCH_VAL(val_dtype, V, {
      SPAMD_DISPATCH_IDX(idx_dtype, I, { BMK_GO(V, I, BmkSplitItems<V>::value, BMK_SPLIT_THREADS, BMK_SPLIT_DUP, true) })
    })
    return SPAMD_ETYPE;
  }
  SPAMD_DISPATCH_VAL(val_dtype, V, {
    SPAMD_DISPATCH_IDX(idx_dtype, I, { BMK_GO(V, I, BmkItems<V>::value, BMK_THREADS, BMK_DUP, false) })
  })
#undef BMK_GO
  return SPAMD_ETYPE;
}
