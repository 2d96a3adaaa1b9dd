// A9 in column-panel order (spamd_sddmm_panels): out[perm[n]] = s_p[n] * <A[rows_p[n], :], Bt[cols_p[n], :]>.
//
// The element order walks the mask one panel of Bt rows at a time (row-major inside a panel; XCD-major when `xstate` is
// given), so that a panel's Bt rows (~3 MB) are L2 hits.  What the first panel kernel (round 2; sddmm.hip, still used for rows
// of 1 KB and more) left on the
// table, by its counters at BASELINE config 4 (profiles/r02_sddmm_sampled_pmc.json): every workgroup lives ~12 us, almost
// all of it waiting on a CHAIN of dependent memory latencies - the mask's arrays, then per batch of four elements the Bt
// rows and, whenever the row changes (every ~6 elements inside a panel), the A row, loaded on the spot and waited for;
// and the A rows streaming through the L2 (one pass over A per panel) evicted the panel's Bt rows (fabric reads 2.2 GB
// against 1.1 GB of A + Bt + mask).  This kernel:
//   * a workgroup takes 256 consecutive elements; thread t loads element t's row, column, value and position (coalesced,
//     non-temporal: a once-through stream);
//   * the DISTINCT A rows of those elements (~41 at config 4: runs of equal rows are found with one block scan) are staged
//     in LDS by ONE burst of loads issued by all 256 threads - no A latency is left on any element's critical path,
//     a row is fetched once per workgroup instead of once per lane group that meets it;
//   * the arithmetic is the row-cached kernel's: the same lanes hold the same K positions, the products are added in
//     the same order, the 16-lane sum is the same DPP tree - results are bit-identical to spamd_sddmm's
//     (tests/test_sddmm_gpu.py::test_sddmm_column_panel_order_is_bit_identical).
#include "sddmm_common.h"
#include <algorithm>
#include <stdlib.h>


namespace spamd {

#ifndef SDP_UNR
#define SDP_UNR 4   // Bt rows per lane group in flight (8: 100 registers, four waves per SIMD, 0.323 ms at config 4; 4: 0.313; 2: 0.316)
#endif
#ifndef SDP_BLK
#define SDP_BLK 256   // elements (= threads) per workgroup
#endif
#ifndef SDP_CHUNKS
#define SDP_CHUNKS 1   // chunks of SDP_BLK elements per workgroup (the next chunk's mask arrays are prefetched)
#endif
#ifndef SDP_NT_A
#define SDP_NT_A 0   // 1: A rows with the non-temporal hint (they then come from HBM instead of the Infinity Cache)
#endif

// 16 bytes per lane from global memory straight into LDS at lds_base + 16 * lane (LDS-DMA; lds_base wave-uniform)
__device__ __forceinline__ void sdp_dma16(unsigned lds_base, const void* src) {
#if SDP_NT_A
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt"
               :
               : "s"(lds_base), "v"(src)
               : "memory", "m0");
#else
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
               :
               : "s"(lds_base), "v"(src)
               : "memory", "m0");
#endif
}

// barrier that orders LDS accesses only: global loads already in flight (the first batch's Bt rows) stay in flight
__device__ __forceinline__ void sdp_lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <typename TIN, typename I, int LPN, int KS, int UNR, int U0>
struct SdpBatch {
  using ACC = typename Acc<TIN>::type;
  using VT = Vec<TIN, 16 / (int)sizeof(TIN)>;
  static constexpr int ROWB = LPN * KS * 16;

  template <int K>
  static __device__ __forceinline__ void load(VT (&bv)[UNR][KS], int (&sl)[UNR], I (&rr)[UNR], I myrow, I mycol, int myslot,
                                              const char* Bb, int64_t ldb_b, int koff_b) {
    if constexpr (K < UNR) {
      const I c = sd_bcast<LPN, U0 + K>(mycol);
      sl[K] = sd_bcast<LPN, U0 + K>(myslot);
      rr[K] = sd_bcast<LPN, U0 + K>(myrow);
      const char* bp = Bb + ((int64_t)c * ldb_b + koff_b);
#pragma unroll
      for (int s = 0; s < KS; ++s) bv[K][s] = *reinterpret_cast<const VT*>(bp + s * (LPN * 16));
      load<K + 1>(bv, sl, rr, myrow, mycol, myslot, Bb, ldb_b, koff_b);
    }
  }

  template <int K>
  static __device__ __forceinline__ void dot(int cnt, int sub, const VT (&bv)[UNR][KS], const int (&sl)[UNR], const I (&rr)[UNR],
                                             const char* sa, int cap, const char* Ab, int64_t lda_b, int koff_b, ACC& res) {
    if constexpr (K < UNR) {
      if (U0 + K < cnt) {
        VT av[KS];
        if (sl[K] < cap) {
          const char* ap = sa + sl[K] * ROWB + koff_b;
#pragma unroll
          for (int s = 0; s < KS; ++s) av[s] = *reinterpret_cast<const VT*>(ap + s * (LPN * 16));
        } else {   // more distinct rows in this workgroup than LDS slots: straight from memory, waited for on the spot
          const char* ap = Ab + ((int64_t)rr[K] * lda_b + koff_b);
#pragma unroll
          for (int s = 0; s < KS; ++s) av[s] = sd_load_now<VT>(ap + s * (LPN * 16));
        }
        const ACC t = sd_dot_group<TIN, VT, LPN, KS>(av, bv[K]);
        res = sub == U0 + K ? t : res;
      }
      dot<K + 1>(cnt, sub, bv, sl, rr, sa, cap, Ab, lda_b, koff_b, res);
    }
  }
};

template <typename TIN, typename I, int LPN, int KS, int UNR, int U0>
struct SdpStep {
  using ACC = typename Acc<TIN>::type;
  using VT = Vec<TIN, 16 / (int)sizeof(TIN)>;
  static __device__ __forceinline__ void run(int cnt, int sub, I myrow, I mycol, int myslot, const char* sa, int cap,
                                             const char* Ab, const char* Bb, int64_t lda_b, int64_t ldb_b, int koff_b, ACC& res) {
    if constexpr (U0 < LPN) {
      if (U0 < cnt) {
        VT bv[UNR][KS];
        int sl[UNR];
        I rr[UNR];
        SdpBatch<TIN, I, LPN, KS, UNR, U0>::template load<0>(bv, sl, rr, myrow, mycol, myslot, Bb, ldb_b, koff_b);
        SdpBatch<TIN, I, LPN, KS, UNR, U0>::template dot<0>(cnt, sub, bv, sl, rr, sa, cap, Ab, lda_b, koff_b, res);
      }
      SdpStep<TIN, I, LPN, KS, UNR, U0 + UNR>::run(cnt, sub, myrow, mycol, myslot, sa, cap, Ab, Bb, lda_b, ldb_b, koff_b, res);
    }
  }
};

// The same steps software-pipelined: the Bt rows of batch U0 + UNR are requested BEFORE batch U0 (whose rows arrived in `bv`)
// is multiplied, so a lane group always has one batch of loads in flight behind the one it works on.  Two batches of
// registers instead of one: the kernel stays below the 96 registers that its LDS-limited five waves per SIMD allow.
template <typename TIN, typename I, int LPN, int KS, int UNR, int U0>
struct SdpPipe {
  using ACC = typename Acc<TIN>::type;
  using VT = Vec<TIN, 16 / (int)sizeof(TIN)>;
  static __device__ __forceinline__ void run(int cnt, int sub, I myrow, I mycol, int myslot, const VT (&bv)[UNR][KS],
                                             const int (&sl)[UNR], const I (&rr)[UNR], const char* sa, int cap, const char* Ab,
                                             const char* Bb, int64_t lda_b, int64_t ldb_b, int koff_b, ACC& res) {
    if constexpr (U0 < LPN) {
      VT nbv[UNR][KS];
      int nsl[UNR];
      I nrr[UNR];
      if constexpr (U0 + UNR < LPN) {
        if (U0 + UNR < cnt)
          SdpBatch<TIN, I, LPN, KS, UNR, U0 + UNR>::template load<0>(nbv, nsl, nrr, myrow, mycol, myslot, Bb, ldb_b, koff_b);
      }
      if (U0 < cnt) SdpBatch<TIN, I, LPN, KS, UNR, U0>::template dot<0>(cnt, sub, bv, sl, rr, sa, cap, Ab, lda_b, koff_b, res);
      if constexpr (U0 + UNR < LPN)
        SdpPipe<TIN, I, LPN, KS, UNR, U0 + UNR>::run(cnt, sub, myrow, mycol, myslot, nbv, nsl, nrr, sa, cap, Ab, Bb, lda_b, ldb_b,
                                                      koff_b, res);
    }
  }
};

#ifndef SDP_PIPE
#define SDP_PIPE 0   // 1: batches software-pipelined (SdpPipe; round 4: 0.374 ms at 104 registers = four waves per SIMD, 0.565 ms
                     // squeezed into 96 with 27 spills, against 0.348 ms for 0: one batch at a time, 60 registers, five waves)
#endif

// BLK threads = BLK consecutive elements per workgroup
#ifndef SDP_WPE
#define SDP_WPE 5   // waves per SIMD the register allocation aims at (five 256-thread workgroups per CU is what the LDS allows)
#endif
template <typename TIN, typename TS, typename I, int LPN, int KS, int UNR, int BLK>
__global__ void __launch_bounds__(BLK) __attribute__((amdgpu_waves_per_eu(SDP_WPE, 8)))
sddmm_panel_kernel(int64_t nnz, int cap, const I* __restrict__ rows, const I* __restrict__ cols,
                   const TS* __restrict__ s_data, const TIN* __restrict__ A, int64_t lda, const TIN* __restrict__ Bt,
                   int64_t ldb, TS* __restrict__ out, const int64_t* __restrict__ perm, const int64_t* __restrict__ xstate,
                   int mode, typename Acc<TIN>::type* __restrict__ part) {
  // mode (rows of 1 KB in two 512-byte halves, see spamd_sddmm_panels): 0 = the whole dot product, out[perm[n]] = s * dot;
  // 1 = first half: part[n] = dot (panel order: coalesced); 2 = last half: out[perm[n]] = s * (part[n] + dot)
  using ACC = typename Acc<TIN>::type;
  using VT = Vec<TIN, 16 / (int)sizeof(TIN)>;
  static_assert(LPN % UNR == 0 && BLK % LPN == 0 && BLK % 64 == 0, "whole batches per step, whole lane groups per workgroup");
  constexpr int ROWB = LPN * KS * 16;   // bytes of a row of A / Bt
  constexpr int VPR = LPN * KS;         // 16-byte vectors per row
  // the only LDS object (starts at LDS byte 0: the LDS-DMA below addresses it through M0): `cap` staged rows of A,
  // then the elements' rows, the distinct rows and the scan's wave totals
  extern __shared__ __attribute__((aligned(16))) char sa[];
  I* const srow = reinterpret_cast<I*>(sa + (size_t)cap * ROWB + 1024);
  I* const drow = srow + BLK;
  int* const wtot = reinterpret_cast<int*>(drow + BLK);
  const int tid = threadIdx.x;
  // this workgroup's SDP_CHUNKS x BLK consecutive elements, BLK at a time.  XCD-private panels (xstate = first[9]): the order
  // is XCD-major and workgroup b takes piece b / 8 of the range of XCD b % 8 - the XCD it is observed to run on; only speed
  // depends on that.  The mask arrays of chunk c + 1 are requested before chunk c is computed (one HBM latency less on the
  // chain of every chunk but the first).
  int64_t wb, we;
  if (xstate) {
    const int x = (int)(blockIdx.x & 7u);
    const int64_t lo = xstate[x], hi = xstate[x + 1];
    wb = lo + (int64_t)(blockIdx.x >> 3) * (BLK * SDP_CHUNKS);
    we = wb + BLK * SDP_CHUNKS < hi ? wb + BLK * SDP_CHUNKS : hi;
  } else {
    wb = (int64_t)blockIdx.x * (BLK * SDP_CHUNKS);
    we = wb + BLK * SDP_CHUNKS < nnz ? wb + BLK * SDP_CHUNKS : nnz;
  }
  if (wb >= we) return;
  const char* const Ab = reinterpret_cast<const char*>(A);
  const char* const Bb = reinterpret_cast<const char*>(Bt);
  const int64_t lda_b = lda * (int64_t)sizeof(TIN), ldb_b = ldb * (int64_t)sizeof(TIN);
  const int lane = tid & 63, wv = tid >> 6;
  const int sub = lane % LPN;
  const int grp = tid / LPN;
  const int koff_b = sub * 16;
  auto fetch = [&](int64_t pb, I& r, I& c, TS& sv, int64_t& pos) {
#if defined(SDP_ABL) && SDP_ABL == 3   // timing ablation (wrong results): the mask arrays come from a 64 K-element window (L2 hits)
    const int64_t nl = (pb + tid < we ? pb + tid : we - 1) & 0xffff;
#else
    const int64_t nl = pb + tid < we ? pb + tid : we - 1;
#endif
    r = __builtin_nontemporal_load(rows + nl);
#if defined(SDP_ABL) && SDP_ABL == 2   // timing ablation (wrong results): every Bt row comes from a 16-row set (no L2 gather)
    c = __builtin_nontemporal_load(cols + nl) & 15;
#else
    c = __builtin_nontemporal_load(cols + nl);
#endif
    if (mode != 1) {   // (the first half stores its partial sums in panel order: neither the mask value nor the position)
      sv = __builtin_nontemporal_load(s_data + nl);
      pos = __builtin_nontemporal_load(perm + nl);
    }
  };
  I nrow, ncol;
  TS ns = TS(0);
  int64_t npos = 0;
  fetch(wb, nrow, ncol, ns, npos);
#pragma unroll 1
  for (int64_t pb = wb; pb < we; pb += BLK) {
    const int64_t pe = pb + BLK < we ? pb + BLK : we;
    const int nblk = (int)(pe - pb);
    const bool mine = tid < nblk;
    const int64_t nl = pb + (mine ? tid : 0);
    const I myrow = nrow, mycol = ncol;
    const TS mys = ns;
    const int64_t mypos = npos;
    if (SDP_CHUNKS > 1 && pe < we) fetch(pe, nrow, ncol, ns, npos);
    ACC prev = 0;      // the first half's sum of my element (requested here, used after the dot product)
    if (mode == 2) prev = __builtin_nontemporal_load(part + nl);

    // distinct rows of the chunk's elements: heads of runs of equal rows, numbered by a block scan
    srow[tid] = myrow;
    __syncthreads();
    const int head = (mine && (tid == 0 || srow[tid - 1] != myrow)) ? 1 : 0;
    int incl = head;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int n = __shfl_up(incl, off, 64);
      if (lane >= off) incl += n;
    }
    if (lane == 63) wtot[wv] = incl;
    __syncthreads();
    int wbase = 0, ndist = 0;
#pragma unroll
    for (int w = 0; w < BLK / 64; ++w) {
      if (w < wv) wbase += wtot[w];
      ndist += wtot[w];
    }
    const int myslot = wbase + incl - 1;   // slot of my element's row (>= 0 for every valid element)
    if (head) drow[myslot] = myrow;
    __syncthreads();

    int cnt = nblk - grp * LPN;               // elements of my lane group (uniform inside the group)
    cnt = cnt < 0 ? 0 : (cnt > LPN ? LPN : cnt);
    // The Bt rows of the lane group's first batch are requested first (into registers), then ONE burst of LDS-DMA brings the
    // staged A rows straight into LDS (no staging registers: `global_load_lds_dwordx4`, a wave-instruction moves 1 KB =
    // 64 consecutive 16-byte vectors of the staged area); both are in flight together and waited for once.
    // (Both 512-byte halves of 1 KB rows inside ONE launch - the chunk's A rows staged half by half, the second half's sums
    // added to the first's - was built and measured in round 5: 0.98 ms at config 4 fp32 against 0.90 for the row-cached
    // kernel and 0.79-0.83 for two launches over half-row panels: the second round repeats the staging burst, its wait and
    // two barriers per chunk, and the panels still hold whole rows.  Removed.)
    const int nstage = ndist < cap ? ndist : cap;
    const int nvec = nstage * VPR;   // (>= VPR: the chunk has at least one element)
    using B0 = SdpBatch<TIN, I, LPN, KS, UNR, 0>;
    VT bv0[UNR][KS];
    int sl0[UNR];
    I rr0[UNR];
    B0::template load<0>(bv0, sl0, rr0, myrow, mycol, myslot, Bb, ldb_b, koff_b);
    for (int base = uniform(wv) * 64; base < nvec; base += BLK) {   // wave-uniform trip count; lanes past the end repeat the last vector
      int i = base + lane;
      i = i < nvec ? i : nvec - 1;
#if defined(SDP_ABL) && SDP_ABL == 4   // timing ablation (wrong results): the staged A rows come from a 1024-row set
      const char* ap = Ab + ((int64_t)(drow[i / VPR] & 1023) * lda_b + (int64_t)(i % VPR) * 16);
#else
      const char* ap = Ab + ((int64_t)drow[i / VPR] * lda_b + (int64_t)(i % VPR) * 16);
#endif
      sdp_dma16((unsigned)uniform(base) * 16u, ap);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    sdp_lds_barrier();

    ACC res = 0;
#if SDP_PIPE
    SdpPipe<TIN, I, LPN, KS, UNR, 0>::run(cnt, sub, myrow, mycol, myslot, bv0, sl0, rr0, sa, cap, Ab, Bb, lda_b, ldb_b, koff_b, res);
#else
    B0::template dot<0>(cnt, sub, bv0, sl0, rr0, sa, cap, Ab, lda_b, koff_b, res);
    SdpStep<TIN, I, LPN, KS, UNR, UNR>::run(cnt, sub, myrow, mycol, myslot, sa, cap, Ab, Bb, lda_b, ldb_b, koff_b, res);
#endif
#if defined(SDP_ABL) && SDP_ABL == 1   // timing ablation (wrong order): results stored in panel order, coalesced
    if (mine) __builtin_nontemporal_store((TS)((ACC)mys * res), out + nl + (mypos & 0));
#else
    if (mode == 1) {
      if (mine) __builtin_nontemporal_store(res, part + nl);
    } else {
      if (mode == 2) res = prev + res;   // (first half + second half: the row-major kernel adds its two halves the same way, sd_dot_1k)
      if (mine) __builtin_nontemporal_store((TS)((ACC)mys * res), out + mypos);  // scattered: keep these lines out of the panel's way
    }
#endif
    if (SDP_CHUNKS > 1) __syncthreads();   // (the next chunk re-uses the staged rows' LDS)
  }
}

template <typename TIN, typename TS, typename I>
static int launch_panel(int64_t nnz, const I* rows, const I* cols, const TS* s, const TIN* A, int64_t lda, const TIN* Bt,
                        int64_t ldb, int64_t K, TS* out, hipStream_t st, const int64_t* perm, int64_t cap_rows,
                        const int64_t* xstate, int64_t xmax, int mode = 0, typename Acc<TIN>::type* part = nullptr) {
  constexpr int EPL = 16 / (int)sizeof(TIN);
  const int64_t vecs = K / EPL;
  for (int L = 16; L <= 64; L <<= 1) {
    if (vecs % L) continue;
    const int ks = (int)(vecs / L);
    if (ks < 1 || ks > 4) continue;
    const int rowb = L * ks * 16;
    // LDS slots for A rows: 24 KB by default (six workgroups per CU), at least 16 rows, never more than a workgroup
    // has elements
    int cap = cap_rows > 0 ? (int)cap_rows : std::max(16, (24 << 10) / rowb);
    const int blk = SDP_BLK;
    if (cap > blk) cap = blk;
    const int room = (int)((160 * 1024 - 1024 - 2 * blk * (int)sizeof(I) - 16 - 2048) / rowb);   // (one workgroup per CU at most)
    if (cap > room) cap = room;
    int64_t blocks = ceil_div(nnz, (int64_t)blk * SDP_CHUNKS);
    if (xstate) blocks = 8 * std::max<int64_t>(ceil_div(xmax, (int64_t)blk * SDP_CHUNKS), 1);
    // (+ 1 KB: the last LDS-DMA instruction of the staging burst always writes a whole KB)
    const size_t lds = (size_t)cap * rowb + 1024 + 2 * blk * sizeof(I) + 16;
#define SDP(LL, KK, UU)                                                                                        \
  if (L == LL && ks == KK) {                                                                                   \
    constexpr int BLK = SDP_BLK;                                                                               \
    auto kern = &sddmm_panel_kernel<TIN, TS, I, LL, KK, UU, BLK>;                                              \
    if (lds > 48 * 1024) {                                                                                     \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                  \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                \
      if (e != hipSuccess) return (int)e;                                                                      \
    }                                                                                                          \
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(BLK), lds, st, nnz, cap, rows, cols, s, A, lda, Bt,  \
                       ldb, out, perm, xstate, mode, part);                                                    \
    return launch_status();                                                                                    \
  }
    SDP(16, 1, SDP_UNR) SDP(16, 2, SDP_UNR) SDP(16, 3, SDP_UNR) SDP(32, 1, SDP_UNR)   // (rows below 1 KB: see spamd_sddmm_panels; 768-byte rows since round 6)
#undef SDP
  }
  return SPAMD_EINVAL;
}

// rows of 1 KB: the first 512-byte halves into `part` (panel order), then the second halves on top of them
template <typename TIN, typename TS, typename I>
static int launch_panel_halves(int64_t nnz, const I* rows, const I* cols, const TS* s, const TIN* A, int64_t lda, const TIN* Bt,
                               int64_t ldb, int64_t K, TS* out, hipStream_t st, const int64_t* perm, int64_t cap_rows,
                               const int64_t* xstate, int64_t xmax, typename Acc<TIN>::type* part) {
  const int64_t kh = K / 2;
  if (int rc = launch_panel<TIN, TS, I>(nnz, rows, cols, s, A, lda, Bt, ldb, kh, out, st, perm, cap_rows, xstate, xmax, 1, part))
    return rc;
  return launch_panel<TIN, TS, I>(nnz, rows, cols, s, A + kh, lda, Bt + kh, ldb, kh, out, st, perm, cap_rows, xstate, xmax, 2, part);
}

}  // namespace spamd

using namespace spamd;

int sddmm_rowcache_panels(int in_dtype, int s_dtype, int idx_dtype, int64_t nnz, const void* rows, const void* cols,
                          const void* s_data, const void* A, int64_t lda, const void* Bt, int64_t ldb, int64_t K, void* out,
                          void* stream, const int64_t* perm, int64_t perm_chunk, const int64_t* xstate, int64_t xmax);

// Column-panel order: rows_p/cols_p/s_p are the mask's coordinates and values gathered by `perm` (the stable sort of
// spamd_sddmm_panel_keys); out stays in the mask's own order: out[perm[n]] = s_p[n] * <A[rows_p[n]], Bt[cols_p[n]]>.
// `chunk` > 0: LDS slots for A rows per workgroup (default: 24 KB worth; rows below 1 KB only).  SPAMD_EINVAL when K has no row-cached kernel
// (use spamd_sddmm).
// Rows of exactly 1 KB (fp32 K = 256, fp64 K = 128, bf16 K = 512) with `part` (nnz accumulator-type words of scratch): TWO
// passes over the mask, one per 512-byte half of the rows (round 5).  The distinct A rows of a workgroup do not fit LDS at
// 1 KB, so until round 4 these rows kept the row-cached kernel, whose panels hold whole 1 KB Bt rows: 32 panels at config 4,
// every one streaming all of A (3.2 GB) - 6.4 GB of fabric traffic for 0.26 GB of operands.  Half-rows double the rows
// a panel holds in the same L2 bytes (the caller builds the panels for 512-byte rows: `spamd_sddmm_panel_row_bytes`), so A
// is streamed 16 x 2 halves = half as often, and each pass is the LDS-staged kernel at its native row length.  Pass 1
// leaves every element's first-half sum in `part` IN PANEL ORDER (coalesced, 4 or 8 bytes per element), pass 2 adds the
// second half and writes s * sum to the element's place.  spamd_sddmm's row-major kernel adds its halves in the same
// order (sd_dot_1k), so the two orders stay bit-identical.
extern "C" int64_t spamd_sddmm_panel_row_bytes(int in_dtype, int64_t K) {
  const int esz = in_dtype == SPAMD_BF16 ? 2 : (in_dtype == SPAMD_F32 ? 4 : (in_dtype == SPAMD_F64 ? 8 : 0));
  if (!esz || K <= 0) return 0;
  return K * esz == 1024 ? 512 : K * esz;
}

extern "C" int spamd_sddmm_panels(int in_dtype, int s_dtype, int idx_dtype, int64_t nnz, const void* rows_p,
                                  const void* cols_p, const int64_t* perm, const void* s_p, const void* A, int64_t lda,
                                  const void* Bt, int64_t ldb, int64_t K, int64_t chunk, const int64_t* xcd_first,
                                  int64_t xcd_max, void* part, void* out, void* stream) {
  if (!perm || (xcd_first && xcd_max < 0) || nnz < 0 || K <= 0) return SPAMD_EINVAL;
  if (nnz == 0) return 0;
  if (((uintptr_t)A % 16) || ((uintptr_t)Bt % 16)) return SPAMD_EINVAL;
  const int esz = in_dtype == SPAMD_BF16 ? 2 : (in_dtype == SPAMD_F32 ? 4 : (in_dtype == SPAMD_F64 ? 8 : 0));
  if (!esz) return SPAMD_ETYPE;
  if ((lda * esz) % 16 || (ldb * esz) % 16 || (K * esz) % 16) return SPAMD_EINVAL;
  // Rows of 1 KB and more (fp32 K = 256, ...): the distinct A rows of a workgroup no longer fit LDS at an occupancy that
  // pays (10 KB per wave; measured 0.92 ms against 0.87 ms at config 4's shapes in fp32), so those keep the row-cached
  // kernel of sddmm.hip, which holds the current A row in registers.
  const bool halves = K * esz == 1024 && part != nullptr;
  if (K * esz >= 1024 && !halves)
    return sddmm_rowcache_panels(in_dtype, s_dtype, idx_dtype, nnz, rows_p, cols_p, s_p, A, lda, Bt, ldb, K, out, stream, perm,
                                 0, xcd_first, xcd_max);
  hipStream_t st = (hipStream_t)stream;
  if (halves) {
    SPAMD_DISPATCH_IDX(idx_dtype, I, {
      const I* r = (const I*)rows_p;
      const I* c = (const I*)cols_p;
      if (in_dtype == SPAMD_BF16 && s_dtype == SPAMD_F32)
        return launch_panel_halves<__hip_bfloat16, float, I>(nnz, r, c, (const float*)s_p, (const __hip_bfloat16*)A, lda,
                                                             (const __hip_bfloat16*)Bt, ldb, K, (float*)out, st, perm, chunk,
                                                             xcd_first, xcd_max, (float*)part);
      if (in_dtype == SPAMD_F32 && s_dtype == SPAMD_F32)
        return launch_panel_halves<float, float, I>(nnz, r, c, (const float*)s_p, (const float*)A, lda, (const float*)Bt, ldb, K,
                                                    (float*)out, st, perm, chunk, xcd_first, xcd_max, (float*)part);
      if (in_dtype == SPAMD_F64 && s_dtype == SPAMD_F64)
        return launch_panel_halves<double, double, I>(nnz, r, c, (const double*)s_p, (const double*)A, lda, (const double*)Bt, ldb,
                                                      K, (double*)out, st, perm, chunk, xcd_first, xcd_max, (double*)part);
    })
    return SPAMD_ETYPE;
  }
  SPAMD_DISPATCH_IDX(idx_dtype, I, {
    const I* r = (const I*)rows_p;
    const I* c = (const I*)cols_p;
    if (in_dtype == SPAMD_BF16 && s_dtype == SPAMD_F32)
      return launch_panel<__hip_bfloat16, float, I>(nnz, r, c, (const float*)s_p, (const __hip_bfloat16*)A, lda,
                                                    (const __hip_bfloat16*)Bt, ldb, K, (float*)out, st, perm, chunk, xcd_first, xcd_max);
    if (in_dtype == SPAMD_F32 && s_dtype == SPAMD_F32)
      return launch_panel<float, float, I>(nnz, r, c, (const float*)s_p, (const float*)A, lda, (const float*)Bt, ldb, K,
                                           (float*)out, st, perm, chunk, xcd_first, xcd_max);
    if (in_dtype == SPAMD_F64 && s_dtype == SPAMD_F64)
      return launch_panel<double, double, I>(nnz, r, c, (const double*)s_p, (const double*)A, lda, (const double*)Bt, ldb, K,
                                             (double*)out, st, perm, chunk, xcd_first, xcd_max);
  })
  return SPAMD_ETYPE;
}
