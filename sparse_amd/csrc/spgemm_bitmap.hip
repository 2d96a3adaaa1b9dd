// A4 / A5, row-local form for WIDE rows of a matrix with at most 2^20 columns (round 4): C = A @ B with both operands
// compressed by rows (reference `_csr_csr_count_nnz` + `_dot_csr_csr`, sparse/numba_backend/_common.py:543-570,639-717).
//
// spgemm_rows.hip orders a row's products with LDS bucket counts + per-bucket sorting networks, writes the row to a
// scratch area at its product offset and needs a pack kernel once every row length is known: ~20 barrier-separated phases
// per row, two passes over the column range, 69 GB of traffic for 20 GB of algorithmic bytes at BASELINE config 5.
// Here the column ORDER comes from a bitmap instead of a sort, and rows go straight to their final place:
//   * one persistent 1024-thread workgroup per CU keeps a bitmap of the output row's columns in LDS (n_col bits: 125 KB
//     at config 5).  Every product sets its column's bit with one returning LDS atomic; the position of a column in the
//     sorted row is the number of set bits below it (a popcount scan over the bitmap + three LDS reads per product), so
//     nothing is sorted and nothing is ranked against anything else;
//   * the first product to set a bit stores (column, value) at its final position.  Products that find their bit already
//     set (two products of one output element: ~50 of 10^4 at config 5) are parked in a small LDS list together with the
//     first arriver of their column (which recognises itself through a 2048-bit filter); the entry with the smallest
//     A-element index of each column then adds the column's products left to right in A's order - the reference's
//     `sums[j] += ...` order (`_common.py:690-705`), bit-identical to spgemm_rows.hip and to the global form;
//   * the row length is known right after the popcount scan, BEFORE anything is emitted: rows take tickets in order, a
//     decoupled look-back over one state word per row (common.h) gives the row's offset in the result, and the row is
//     written once, in place.  No scratch rows, no pack kernel, no scan over the row lengths; the exact zeros written are
//     counted on the way (the result container's prune then has nothing to read);
//   * a row's products are prefetched: while row r is ranked and emitted, the loads of row r+1's products are in flight
//     (into registers) and row r+2's A elements are being fetched, so no phase waits for HBM.  Barriers wait for LDS
//     only (`s_waitcnt lgkmcnt(0)` + `s_barrier`): `__syncthreads()` would drain the prefetch at every phase.
// Limits (checked by the host before the launch, spamd_spgemm_bitmap_limits): n_col <= 2^20, A rows of at most 256
// elements, at most 1024 * ITEMS products per row, index arrays of either width.  A row whose parked products exceed the
// list (512 entries) sets the `failed` word: the caller then discards the result and uses spgemm_rows.hip.
#include <mutex>

#include "common.h"

namespace spamd {

constexpr int BMK_THREADS = 1024;
constexpr int BMK_STAGE = 256;                              // A elements a row may have
constexpr int BMK_MAX_GROUPS = 4096;                        // groups of 256 columns (8 bitmap words)
constexpr int BMK_GPT = BMK_MAX_GROUPS / BMK_THREADS;       // groups per thread = 16-bit fields of the packed scan
constexpr int BMK_DUP = 512;                                // parked products per row
constexpr int BMK_FILT_WORDS = 64;                          // 2048-bit filter of the columns with parked products
constexpr unsigned BMK_NONE = 0xffffffffu;

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename V>
struct BmkDup {
  unsigned key;   // (column << 8) | index of the A element
  unsigned rank;  // position of the column in the row (first arrivers), BMK_NONE otherwise
  V val;
};

template <typename V>
struct BmkStage {   // one A row: prefix[e] = products of the elements before e, start of B row k_e, A value
  int prefix[BMK_STAGE + 4];
  int64_t bstart[BMK_STAGE];
  V aval[BMK_STAGE];
};

struct BmkMisc {
  unsigned long long wa[20];
  unsigned wb[20];
  int ndup;
  int pad;
  int64_t row_off;
  int64_t ticket[2];
};

template <typename V>
struct BmkLayout {
  __host__ __device__ static size_t bitmap_bytes(int ngroups) { return (size_t)ngroups * 32; }
  __host__ __device__ static size_t pref_bytes(int ngroups) { return ((size_t)ngroups * 2 + 15) / 16 * 16; }
  static size_t bytes(int ngroups) {
    return bitmap_bytes(ngroups) + pref_bytes(ngroups) + sizeof(BmkDup<V>) * BMK_DUP + BMK_FILT_WORDS * 4 +
           2 * sizeof(BmkStage<V>) + sizeof(BmkMisc) + 64;
  }
};

// exclusive block scan of (a: four packed 16-bit counts, b: one 32-bit count); totals in ta / tb.  Two LDS barriers.
__device__ __forceinline__ void bmk_block_scan(unsigned long long& a, unsigned& b, unsigned long long& ta, unsigned& tb,
                                               BmkMisc* m) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  unsigned long long xa = a;
  unsigned xb = b;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long ya = __shfl_up(xa, d, 64);
    const unsigned yb = __shfl_up(xb, d, 64);
    if (lane >= d) {
      xa += ya;
      xb += yb;
    }
  }
  if (lane == 63) {
    m->wa[wid] = xa;
    m->wb[wid] = xb;
  }
  lds_barrier();
  if (wid == 0) {
    constexpr int NW = BMK_THREADS / 64;
    const unsigned long long wa = lane < NW ? m->wa[lane] : 0;
    const unsigned wb = lane < NW ? m->wb[lane] : 0;
    unsigned long long sa = wa;
    unsigned sb = wb;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
      const unsigned long long ya = __shfl_up(sa, d, 64);
      const unsigned yb = __shfl_up(sb, d, 64);
      if (lane >= d) {
        sa += ya;
        sb += yb;
      }
    }
    if (lane < NW) {
      m->wa[lane] = sa - wa;
      m->wb[lane] = sb - wb;
    }
    if (lane == NW - 1) {
      m->wa[NW] = sa;
      m->wb[NW] = sb;
    }
  }
  lds_barrier();
  a = xa - a + m->wa[wid];
  b = xb - b + m->wb[wid];
  ta = m->wa[BMK_THREADS / 64];
  tb = m->wb[BMK_THREADS / 64];
}

__device__ __forceinline__ int bmk_popc_group(const unsigned* bm, int g) {
  const uint4 lo = *reinterpret_cast<const uint4*>(bm + (size_t)g * 8);
  const uint4 hi = *reinterpret_cast<const uint4*>(bm + (size_t)g * 8 + 4);
  return __popc(lo.x) + __popc(lo.y) + __popc(lo.z) + __popc(lo.w) + __popc(hi.x) + __popc(hi.y) + __popc(hi.z) + __popc(hi.w);
}

// number of set bits below column `col`: the column's position in the sorted row
__device__ __forceinline__ int bmk_rank(const unsigned* bm, const unsigned short* pref, unsigned col) {
  const unsigned g = col >> 8, wi = (col >> 5) & 7u;
  const uint4 lo = *reinterpret_cast<const uint4*>(bm + (size_t)g * 8);
  const uint4 hi = *reinterpret_cast<const uint4*>(bm + (size_t)g * 8 + 4);
  const unsigned w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  int r = (int)pref[g];
  unsigned cur = w[0];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    r += (unsigned)i < wi ? __popc(w[i]) : 0;
    cur = (unsigned)i == wi ? w[i] : cur;
  }
  return r + __popc(cur & ((1u << (col & 31u)) - 1u));
}

template <typename V>
__device__ __forceinline__ int bmk_is_zero_bits(V v) {
  if constexpr (sizeof(V) == 8) return __builtin_bit_cast(unsigned long long, v) == 0;
  else return __builtin_bit_cast(unsigned, v) == 0;
}

template <typename V, typename I, int ITEMS>
__global__ void __launch_bounds__(BMK_THREADS)
spgemm_bitmap_kernel(int64_t n_row, int ngroups, const I* __restrict__ a_ptr, const I* __restrict__ a_idx,
                     const V* __restrict__ a_val, const I* __restrict__ b_ptr, const I* __restrict__ b_idx,
                     const V* __restrict__ b_val, unsigned long long* __restrict__ work, int64_t* __restrict__ out_ptr,
                     int64_t* __restrict__ out_idx, V* __restrict__ out_val) {
#pragma clang fp contract(off)
  static_assert(ITEMS % 4 == 0, "A-element indices are packed four to a register");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned* const bm = reinterpret_cast<unsigned*>(smem);
  unsigned short* const pref = reinterpret_cast<unsigned short*>(smem + BmkLayout<V>::bitmap_bytes(ngroups));
  BmkDup<V>* const dup = reinterpret_cast<BmkDup<V>*>(reinterpret_cast<char*>(pref) + BmkLayout<V>::pref_bytes(ngroups));
  unsigned* const filt = reinterpret_cast<unsigned*>(dup + BMK_DUP);
  BmkStage<V>* const stage = reinterpret_cast<BmkStage<V>*>(filt + BMK_FILT_WORDS);
  BmkMisc* const misc = reinterpret_cast<BmkMisc*>(stage + 2);
  unsigned long long* const ticket_ctr = work;
  unsigned long long* const state = work + 8;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  constexpr int CAP = BMK_THREADS * ITEMS;

  // ---- set-up: clean LDS, the first two tickets -------------------------------------------------------------------------
  for (int i = tid; i < ngroups * 2; i += BMK_THREADS) reinterpret_cast<uint4*>(bm)[i] = make_uint4(0, 0, 0, 0);
  if (tid < BMK_FILT_WORDS) filt[tid] = 0;
  if (tid == 0) {
    misc->ndup = 0;
    misc->ticket[0] = (int64_t)__hip_atomic_fetch_add(ticket_ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    misc->ticket[1] = (int64_t)__hip_atomic_fetch_add(ticket_ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  lds_barrier();
  int64_t cur = misc->ticket[0], nxt = misc->ticket[1];
  bool failed = false;

  // the A row of `row`: element `tid` (column of A = row of B, value); rows longer than the staging area fail the call
  auto load_arow = [&](int64_t row, int& nA, int64_t& ka, V& av) {
    nA = 0;
    ka = 0;
    av = V(0);
    if (row < n_row) {
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
  auto load_brow = [&](int nA, int64_t ka, int64_t& bs, unsigned& len) {
    bs = 0;
    len = 0;
    if (tid < nA) {
      bs = (int64_t)b_ptr[ka];
      len = (unsigned)((int64_t)b_ptr[ka + 1] - bs);
    }
  };
  // products of the staged row -> registers (loads only: nothing here waits for them).  Product p belongs to thread
  // p % 1024: for one item index consecutive lanes read consecutive entries of a B row.
  unsigned colN[ITEMS];
  V bvN[ITEMS];
  unsigned eN[ITEMS / 4];
  auto expand = [&](const BmkStage<V>* st, int nA, int P) {
#pragma unroll
    for (int j = 0; j < ITEMS / 4; ++j) eN[j] = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      const int p = j * BMK_THREADS + tid;
      colN[j] = BMK_NONE;
      bvN[j] = V(0);
      // the A element of product p: the last e with prefix[e] <= p (empty B rows are skipped by construction).  A
      // branch-free binary search: data-dependent loops here, sixteen times over, cost the register allocator 300 spills.
      int e = 0;
#pragma unroll
      for (int step = BMK_STAGE / 2; step >= 1; step >>= 1) {
        const int t = e + step;              // (<= 255: always inside the staging array, whatever nA is)
        const int at_t = st->prefix[t];
        e = ((t < nA) & (at_t <= p)) ? t : e;
      }
      if (p < P) {
        const int64_t q = st->bstart[e] + (int64_t)(p - st->prefix[e]);
        colN[j] = (unsigned)b_idx[q];
        bvN[j] = b_val[q];
        eN[j / 4] |= (unsigned)e << (8 * (j % 4));
      }
      if (j % 4 == 3) __builtin_amdgcn_sched_barrier(0);   // (keeps the scheduler from hoisting all 2 x ITEMS address chains)
    }
  };

  // ---- prologue: stage the first row, request its products, fetch the second row's A elements ---------------------------
  int nA_c, nA_n, P_c;
  int64_t ka;
  V av;
  {
    int64_t bs;
    unsigned len;
    load_arow(cur, nA_c, ka, av);
    load_brow(nA_c, ka, bs, len);
    unsigned long long za = 0, ta;
    unsigned excl = len, tl;
    bmk_block_scan(za, excl, ta, tl, misc);
    if (tl > (unsigned)CAP) {
      failed = true;
      tl = 0;
    }
    if (tid <= nA_c) stage[0].prefix[tid] = tl ? (int)excl : 0;
    if (tid < nA_c) {
      stage[0].bstart[tid] = bs;
      stage[0].aval[tid] = av;
    }
    P_c = (int)tl;
    lds_barrier();
    expand(&stage[0], nA_c, P_c);
    load_arow(nxt, nA_n, ka, av);
  }
  int buf = 0;
  int zero_count = 0;

  while (cur < n_row) {   // (workgroup-uniform)
    const BmkStage<V>* const sc = &stage[buf];
    BmkStage<V>* const sn = &stage[buf ^ 1];
    // ---- top: a ticket for the row after next, the B row pointers of the next row's A elements -------------------------
    unsigned long long tk = 0;
    if (tid == 0) tk = __hip_atomic_fetch_add(ticket_ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int64_t bs;
    unsigned len;
    load_brow(nA_n, ka, bs, len);
    const V av_n = av;
    // ---- 1. every product of the current row sets its column's bit -----------------------------------------------------
    // key[j] = (column << 8) | A element; first arrivers are remembered in a mask, later ones parked
    unsigned key[ITEMS];
    V val[ITEMS];
    unsigned first_mask = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      key[j] = BMK_NONE;
      val[j] = V(0);
      if (colN[j] != BMK_NONE) {
        const unsigned e = (eN[j / 4] >> (8 * (j % 4))) & 255u;
        val[j] = sc->aval[e] * bvN[j];
        const unsigned c = colN[j];
        key[j] = (c << 8) | e;
        const unsigned bit = 1u << (c & 31u);
        const unsigned old = atomicOr(&bm[c >> 5], bit);
        if (old & bit) {   // the output element has a product already: park this one
          const unsigned h = (c ^ (c >> 11)) & 2047u;
          atomicOr(&filt[h >> 5], 1u << (h & 31u));
          const int slot = atomicAdd(&misc->ndup, 1);
          if (slot < BMK_DUP) {
            dup[slot].key = key[j];
            dup[slot].rank = BMK_NONE;
            dup[slot].val = val[j];
          }
        } else {
          first_mask |= 1u << j;
        }
      }
      if (j % 4 == 3) __builtin_amdgcn_sched_barrier(0);
    }
    lds_barrier();
    // ---- 2. popcount scan: positions of the columns; the same scan sums the next row's B-row lengths --------------------
    unsigned long long cnts = 0;
#pragma unroll
    for (int m = 0; m < BMK_GPT; ++m) {
      const int g = tid + BMK_THREADS * m;
      if (g < ngroups) cnts |= (unsigned long long)bmk_popc_group(bm, g) << (16 * m);
    }
    unsigned long long excl_c = cnts, tot_c;
    unsigned excl_l = len, tot_l;
    bmk_block_scan(excl_c, excl_l, tot_c, tot_l, misc);
    int row_nnz = 0;
#pragma unroll
    for (int m = 0; m < BMK_GPT; ++m) {
      const int g = tid + BMK_THREADS * m;
      if (g < ngroups) pref[g] = (unsigned short)(row_nnz + (int)((excl_c >> (16 * m)) & 0xffffu));
      row_nnz += (int)((tot_c >> (16 * m)) & 0xffffu);
    }
    if (tot_l > (unsigned)CAP) {
      failed = true;
      tot_l = 0;
    }
    const int P_n = (int)tot_l;
    if (tid <= nA_n) sn->prefix[tid] = P_n ? (int)excl_l : 0;
    if (tid < nA_n) {
      sn->bstart[tid] = bs;
      sn->aval[tid] = av_n;
    }
    if (tid == 0) misc->ticket[0] = (int64_t)tk;
    lds_barrier();
    const int64_t nn = misc->ticket[0];
    // ---- 3. the row's length is known: wave 0 publishes it and looks back for the row's offset, while every first
    // arriver finds its column's position (unless products of its column are parked: then it joins them) ---------------
    if (wid == 0) {
      const unsigned long long before = lookback_exclusive(state, cur, (unsigned long long)row_nnz, lane);
      if (lane == 0) {
        misc->row_off = (int64_t)before;
        out_ptr[cur + 1] = (int64_t)before + row_nnz;
        if (cur == 0) out_ptr[0] = 0;
      }
    }
    // (a position has 14 bits: its low 12 replace the A-element index in key[j], the high 2 of all items share one register)
    unsigned rk_hi = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      if ((first_mask >> j) & 1u) {
        const unsigned c = key[j] >> 8;
        const unsigned h = (c ^ (c >> 11)) & 2047u;
        const bool parked = (filt[h >> 5] >> (h & 31u)) & 1u;
        const int r = bmk_rank(bm, pref, c);
        if (parked) {
          first_mask &= ~(1u << j);
          const int slot = atomicAdd(&misc->ndup, 1);
          if (slot < BMK_DUP) {
            dup[slot].key = key[j];
            dup[slot].rank = (unsigned)r;
            dup[slot].val = val[j];
          }
        }
        key[j] = (c << 12) | ((unsigned)r & 0xfffu);
        rk_hi |= ((unsigned)r >> 12) << (2 * j);
      }
      if (j % 2 == 1) __builtin_amdgcn_sched_barrier(0);   // (two lookups = 18 words in flight, not 16 x 9)
    }
    lds_barrier();
    const int64_t row_off = misc->row_off;
    // ---- 4. request the next row's products (they land while this row is written), then store this row in place ---------
    expand(sn, nA_n, P_n);
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      if ((first_mask >> j) & 1u) {
        const int64_t at = row_off + (int64_t)((key[j] & 0xfffu) | (((rk_hi >> (2 * j)) & 3u) << 12));
        out_idx[at] = (int64_t)(key[j] >> 12);
        out_val[at] = val[j];
        zero_count += bmk_is_zero_bits(val[j]);
      }
      if (j % 4 == 3) __builtin_amdgcn_sched_barrier(0);
    }
    // ---- 5. parked products: the entry with the smallest A-element index of a column sums the column left to right ------
    {
      int n = misc->ndup;
      if (n > BMK_DUP) {
        failed = true;
        n = BMK_DUP;
      }
      if (tid < n) {
        const unsigned kd = dup[tid].key;
        const unsigned c = kd >> 8;
        bool leader = true;
        unsigned rank = dup[tid].rank;
        for (int i = 0; i < n; ++i) {
          const unsigned k = dup[i].key;
          if ((k >> 8) == c) {
            leader = leader && k >= kd;
            const unsigned r2 = dup[i].rank;
            rank = r2 != BMK_NONE ? r2 : rank;
          }
        }
        if (leader) {
          V acc = dup[tid].val;
          unsigned last = kd;
          for (;;) {   // next larger A-element index of this column
            unsigned best = BMK_NONE;
            V bv = V(0);
            for (int i = 0; i < n; ++i) {
              const unsigned k = dup[i].key;
              if ((k >> 8) == c && k > last && k < best) {
                best = k;
                bv = dup[i].val;
              }
            }
            if (best == BMK_NONE) break;
            acc = acc + bv;
            last = best;
          }
          out_idx[row_off + rank] = (int64_t)c;
          out_val[row_off + rank] = acc;
          zero_count += bmk_is_zero_bits(acc);
        }
      }
    }
    // the A elements of the row after next (two dependent loads: they have the rest of this row and the head of the next)
    int nA_nn;
    load_arow(nn, nA_nn, ka, av);
    // ---- 6. leave LDS clean for the next row (nobody reads the bitmap after step 3; the list is read in step 5) ---------
    for (int i = tid; i < ngroups * 2; i += BMK_THREADS) reinterpret_cast<uint4*>(bm)[i] = make_uint4(0, 0, 0, 0);
    if (tid < BMK_FILT_WORDS) filt[tid] = 0;
    lds_barrier();
    if (tid == 0) misc->ndup = 0;
    cur = nxt;
    nxt = nn;
    nA_c = nA_n;
    nA_n = nA_nn;
    P_c = P_n;
    buf ^= 1;
  }
  // ---- epilogue: exact zeros written, failure word ------------------------------------------------------------------------
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) zero_count += __shfl_xor(zero_count, d, 64);
  if (lane == 0 && zero_count) atomicAdd(work + 2, (unsigned long long)zero_count);
  if (failed && lane == 0) __hip_atomic_store(work + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename V>
struct BmkItems {
  static constexpr int value = sizeof(V) == 4 ? 16 : 12;
};

template <typename V, typename I>
static int bmk_launch(int64_t n_row, int64_t n_col, const I* a_ptr, const I* a_idx, const V* a_val, const I* b_ptr,
                      const I* b_idx, const V* b_val, unsigned long long* work, int64_t* out_ptr, int64_t* out_idx, V* out_val,
                      hipStream_t s) {
  constexpr int ITEMS = BmkItems<V>::value;
  auto kern = &spgemm_bitmap_kernel<V, I, ITEMS>;
  const int ngroups = (int)ceil_div(n_col, (int64_t)256);
  const size_t lds = BmkLayout<V>::bytes(ngroups);
  {
    static std::mutex mu;
    static bool done = false;   // (one flag per template instantiation)
    std::lock_guard<std::mutex> lock(mu);
    if (!done) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)BmkLayout<V>::bytes(BMK_MAX_GROUPS));
      if (e != hipSuccess) return (int)e;
      done = true;
    }
  }
  int dev = 0, cus = 0;
  if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return (int)e;
  if (hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); e != hipSuccess) return (int)e;
  // one workgroup per CU: 1024 threads at up to 128 registers fill a CU's register file, whatever the LDS footprint
  const int64_t grid = n_row < cus ? n_row : cus;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(BMK_THREADS), lds, s, n_row, ngroups, a_ptr, a_idx, a_val, b_ptr, b_idx,
                     b_val, work, out_ptr, out_idx, out_val);
  return launch_status();
}

}  // namespace spamd

using namespace spamd;

// limits of the bitmap form: which = 0: products per row, 1: A elements per row, 2: columns, 3: parked products per row
extern "C" int64_t spamd_spgemm_bitmap_limits(int val_dtype, int which) {
  const bool v4 = val_dtype == SPAMD_F32 || val_dtype == SPAMD_I32;
  switch (which) {
    case 0: return (int64_t)BMK_THREADS * (v4 ? BmkItems<float>::value : BmkItems<double>::value);
    case 1: return BMK_STAGE;
    case 2: return (int64_t)BMK_MAX_GROUPS * 256;
    case 3: return BMK_DUP;
    default: return -1;
  }
}

// C = A @ B, rows written in place: out_indptr[n_row + 1], out_indices / out_data with room for every product (the
// caller trims to out_indptr[n_row]).  work: n_row + 8 words, zeroed here; afterwards work[1] != 0 = failed (a row
// outside the limits, or with more parked products than the list holds: discard the result), work[2] = values written
// whose bits are all zero.
extern "C" int spamd_spgemm_bitmap(int val_dtype, int idx_dtype, int64_t n_row, int64_t n_col, const void* a_indptr,
                                   const void* a_indices, const void* a_data, const void* b_indptr, const void* b_indices,
                                   const void* b_data, int64_t* work, int64_t* out_indptr, int64_t* out_indices,
                                   void* out_data, void* stream) {
  if (n_row < 0 || n_col <= 0 || n_col > (int64_t)BMK_MAX_GROUPS * 256 || !work || !out_indptr) return SPAMD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (hipError_t e = hipMemsetAsync(work, 0, (size_t)(n_row + 8) * sizeof(int64_t), s); e != hipSuccess) return (int)e;
  if (n_row == 0) return (int)hipMemsetAsync(out_indptr, 0, sizeof(int64_t), s);
  SPAMD_DISPATCH_VAL(val_dtype, V, {
    SPAMD_DISPATCH_IDX(idx_dtype, I, {
      return (bmk_launch<V, I>(n_row, n_col, (const I*)a_indptr, (const I*)a_indices, (const V*)a_data, (const I*)b_indptr,
                               (const I*)b_indices, (const V*)b_data, reinterpret_cast<unsigned long long*>(work), out_indptr,
                               out_indices, (V*)out_data, s));
    })
  })
  return SPAMD_ETYPE;
}
