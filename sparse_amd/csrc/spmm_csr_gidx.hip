// A1, LDS-tile form with dynamically indexed register accumulators (fp32, N = 128, FMA mode):
// CSR x dense -> dense (reference loop: sparse/numba_backend/_common.py:744-753).
//
// Design (see DESIGN.md section 3/6 for the measurements that led here):
//   * a workgroup of 4 waves owns 256 consecutive rows for the whole kernel; lane l of wave w does
//     the BOOK-KEEPING of row w*64+l, and the wave keeps all 64 rows' partial sums in a fixed VGPR
//     block v[128:255] (row r -> v[128+2r], v[129+2r]; lane l holds columns 2l, 2l+1 of every row);
//   * B streams through LDS in tiles of KB rows (LDS-DMA, double-buffered, one barrier per tile):
//     the vector-L1 traffic for B drops from nnz*512 B to (M/256)*K*512 B;
//   * each row's stored elements are staged, lane <-> element (coalesced LDS-DMA), into a 32-entry
//     LDS window per row; lanes then read THEIR row's next elements from LDS (a transposition that
//     costs no global gather);
//   * per tile, phase A (vectorised over the 64 rows): count the row's elements that fall into the
//     tile, wave-scan the counts, write a flat (row, LDS offset of the B row, value) list;
//     phase B (inline asm): walk the list with 4 ds_read_b64 in flight; the accumulator of the
//     element's row is addressed with s_set_gpr_idx_on (SRC2|DST relative), so there is no
//     per-row code and no wasted slot for rows without elements in the tile.
// Summation order per output element is still storage (k-ascending) order: tiles ascend in k and a
// row's elements keep their order inside the list.  One fused multiply-add per element (the
// bit-exact mul+add mode stays on the row-group kernel).
//
// STATUS (round 1, MI355X, config 2): correct (26 parity cases incl. refills, empty/dense rows, int64
// indices) but 10.7 ms — NOT the default (SPAMD_SPMM_VARIANT="GIDX=1"; "GIDX=2" = debug build with
// memory accumulators; 3/4/5 = timing ablations).  Ablation: the asm consume costs ~5.1 ms (with 256
// VGPRs per wave only 4 waves fit a CU, so every ds_read -> fma chain of the one wave per SIMD is
// exposed: ~75 cycles per element), everything else ~5.6 ms (per-tile fixed work: 8 window reads, a
// 6-step wave scan, list write/read, barrier, ~2000 cycles x 157 tiles x 15 workgroups per CU, plus
// window refills), tile DMA ~1 ms.  What it would take: 2 workgroups per CU (smaller windows/tiles),
// a software pipeline that builds tile t+1's list while tile t is consumed, 8 LDS reads in flight
// across groups, DPP scan.  The dynamic accumulator indexing itself (s_set_gpr_idx) works as intended.
#include "spmm_internal.h"

namespace spamd {

constexpr int GI_WAVES = 4, GI_RW = 64, GI_KB = 64, GI_WIN = 32, GI_WSTRIDE = 33, GI_LIST = 256;
constexpr int GI_TILE = GI_KB * 512;
constexpr int GI_WAVE_LDS = (2 * GI_RW * GI_WSTRIDE + 3 * GI_LIST) * 4;
constexpr int GI_LDS = 2 * GI_TILE + GI_WAVES * GI_WAVE_LDS;
constexpr int GI_SENT = 0x7fffffff;

__device__ __forceinline__ void gi_dma16(unsigned lds_base, const void* src) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
               :
               : "s"(lds_base), "v"(src)
               : "memory", "m0");
}
__device__ __forceinline__ void gi_dma4(unsigned lds_base, const void* src) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off"
               :
               : "s"(lds_base), "v"(src)
               : "memory", "m0");
}

#define GI_ACC_CLOBBERS                                                                                   \
  "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140",   \
      "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", \
      "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", \
      "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", \
      "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", \
      "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", \
      "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", \
      "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", \
      "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", \
      "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"

// ---- phase B: consume `cnt` list entries held lane <-> entry in (er, eo, ev) ------------------------
// er = row (0..63), eo = LDS byte address of the B row, ev = value bits; lane8 = lane*8.
__device__ __forceinline__ void gi_consume_asm(int er, int eo, int ev, int cnt, int lane8) {
  asm volatile(
      "s_mov_b32 s36, 0\n\t"
      "s_sub_i32 s37, %3, 3\n\t"                     // groups of 4 while e < cnt-3
      "s_cmp_lt_i32 s36, s37\n\t"
      "s_cbranch_scc0 2f\n\t"
      "1:\n\t"
      "v_readlane_b32 s40, %1, s36\n\t"
      "s_add_i32 s38, s36, 1\n\t"
      "v_readlane_b32 s41, %1, s38\n\t"
      "s_add_i32 s39, s36, 2\n\t"
      "v_readlane_b32 s42, %1, s39\n\t"
      "s_add_i32 s44, s36, 3\n\t"
      "v_readlane_b32 s43, %1, s44\n\t"
      "v_add_u32 v104, s40, %4\n\t"
      "v_add_u32 v105, s41, %4\n\t"
      "v_add_u32 v106, s42, %4\n\t"
      "v_add_u32 v107, s43, %4\n\t"
      "ds_read_b64 v[96:97], v104\n\t"
      "ds_read_b64 v[98:99], v105\n\t"
      "ds_read_b64 v[100:101], v106\n\t"
      "ds_read_b64 v[102:103], v107\n\t"
      // entry 0
      "v_readlane_b32 s40, %0, s36\n\t"
      "v_readlane_b32 s45, %2, s36\n\t"
      "s_lshl_b32 s40, s40, 1\n\t"
      "s_waitcnt lgkmcnt(3)\n\t"
      "s_set_gpr_idx_on s40, gpr_idx(SRC2,DST)\n\t"
      "v_fma_f32 v128, s45, v96, v128\n\t"
      "v_fma_f32 v129, s45, v97, v129\n\t"
      "s_set_gpr_idx_off\n\t"
      // entry 1
      "v_readlane_b32 s40, %0, s38\n\t"
      "v_readlane_b32 s45, %2, s38\n\t"
      "s_lshl_b32 s40, s40, 1\n\t"
      "s_waitcnt lgkmcnt(2)\n\t"
      "s_set_gpr_idx_on s40, gpr_idx(SRC2,DST)\n\t"
      "v_fma_f32 v128, s45, v98, v128\n\t"
      "v_fma_f32 v129, s45, v99, v129\n\t"
      "s_set_gpr_idx_off\n\t"
      // entry 2
      "v_readlane_b32 s40, %0, s39\n\t"
      "v_readlane_b32 s45, %2, s39\n\t"
      "s_lshl_b32 s40, s40, 1\n\t"
      "s_waitcnt lgkmcnt(1)\n\t"
      "s_set_gpr_idx_on s40, gpr_idx(SRC2,DST)\n\t"
      "v_fma_f32 v128, s45, v100, v128\n\t"
      "v_fma_f32 v129, s45, v101, v129\n\t"
      "s_set_gpr_idx_off\n\t"
      // entry 3
      "v_readlane_b32 s40, %0, s44\n\t"
      "v_readlane_b32 s45, %2, s44\n\t"
      "s_lshl_b32 s40, s40, 1\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_set_gpr_idx_on s40, gpr_idx(SRC2,DST)\n\t"
      "v_fma_f32 v128, s45, v102, v128\n\t"
      "v_fma_f32 v129, s45, v103, v129\n\t"
      "s_set_gpr_idx_off\n\t"
      "s_add_i32 s36, s36, 4\n\t"
      "s_cmp_lt_i32 s36, s37\n\t"
      "s_cbranch_scc1 1b\n\t"
      "2:\n\t"                                       // tail: one entry at a time
      "s_cmp_lt_i32 s36, %3\n\t"
      "s_cbranch_scc0 3f\n\t"
      "v_readlane_b32 s40, %1, s36\n\t"
      "v_readlane_b32 s41, %0, s36\n\t"
      "v_readlane_b32 s45, %2, s36\n\t"
      "v_add_u32 v104, s40, %4\n\t"
      "ds_read_b64 v[96:97], v104\n\t"
      "s_lshl_b32 s41, s41, 1\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_set_gpr_idx_on s41, gpr_idx(SRC2,DST)\n\t"
      "v_fma_f32 v128, s45, v96, v128\n\t"
      "v_fma_f32 v129, s45, v97, v129\n\t"
      "s_set_gpr_idx_off\n\t"
      "s_add_i32 s36, s36, 1\n\t"
      "s_branch 2b\n\t"
      "3:\n\t"
      :
      : "v"(er), "v"(eo), "v"(ev), "s"(cnt), "v"(lane8)
      : "memory", "m0", "scc", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "v96", "v97",
        "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", GI_ACC_CLOBBERS);
}

template <typename I, bool ASM, int DBG>
__global__ void __launch_bounds__(256)
#if 1
__attribute__((amdgpu_num_vgpr(128)))
#endif
spmm_csr_gidx_kernel(int64_t M, int64_t K, const float* __restrict__ a_data, const I* __restrict__ a_idx,
                     const I* __restrict__ a_ptr, const float* __restrict__ b, int64_t ldb,
                     float* __restrict__ out, int64_t ldo) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = uniform(tid >> 6);
  int* const wcol = reinterpret_cast<int*>(lds + 2 * GI_TILE + wv * GI_WAVE_LDS);
  float* const wval = reinterpret_cast<float*>(wcol + GI_RW * GI_WSTRIDE);
  int* const lrow = reinterpret_cast<int*>(wval + GI_RW * GI_WSTRIDE);
  int* const loff = lrow + GI_LIST;
  float* const lval = reinterpret_cast<float*>(loff + GI_LIST);
  const unsigned lds0 = (unsigned)(size_t)lds;

  const int64_t row0 = (int64_t)blockIdx.x * (GI_WAVES * GI_RW) + (int64_t)wv * GI_RW;
  const int64_t row = row0 + lane;
  int64_t p = 0, pend = 0;  // my row's cursor / end (absolute element positions)
  if (row < M) {
    p = (int64_t)a_ptr[row];
    pend = (int64_t)a_ptr[row + 1];
  }
  int64_t wbase = p;  // the LDS window of my row holds elements [wbase, wbase + 32)

  float accs[ASM ? 1 : 2 * GI_RW];  // debug build: accumulators in (scratch) memory
  if constexpr (ASM) {
    asm volatile(
        ".set spamd_gi, 128\n\t"
        ".rept 128\n\t"
        "v_mov_b32 v[spamd_gi], 0\n\t"
        ".set spamd_gi, spamd_gi+1\n\t"
        ".endr\n\t" ::
            : "memory", GI_ACC_CLOBBERS);
  } else {
    for (int i = 0; i < 2 * GI_RW; ++i) accs[i] = 0.f;
  }

  // ---- (re)fill every row's window with its next 32 elements: one LDS-DMA per row and array ----------
  auto refill = [&]() {
    for (int r = 0; r < GI_RW; ++r) {
      const int64_t pr = wave_bcast(p, r);
      const int64_t er = wave_bcast(pend, r);
      const int64_t n = (er - pr) < GI_WIN ? (er - pr) : (int64_t)GI_WIN;
      if (lane < n) {
        gi_dma4((unsigned)(size_t)(wcol + r * GI_WSTRIDE), reinterpret_cast<const char*>(a_idx + pr + lane));
        gi_dma4((unsigned)(size_t)(wval + r * GI_WSTRIDE), a_data + pr + lane);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wbase = p;
  };

  const int64_t ntiles = (K + GI_KB - 1) / GI_KB;
  auto issue_tile = [&](int64_t t) {
    const int64_t kb0 = t * GI_KB;
    const unsigned buf = (unsigned)(t & 1) * GI_TILE;
#pragma unroll
    for (int i = 0; i < GI_TILE / 16 / 256; ++i) {
      const int e = (i * 256 + tid) * 4;  // float index inside the tile
      const int r = e >> 7, c = e & 127;
      if (kb0 + r < K) gi_dma16(lds0 + buf + (unsigned)(i * GI_WAVES + wv) * 1024u, b + (kb0 + r) * ldb + c);
    }
  };

  issue_tile(0);
  refill();  // ends with vmcnt(0): tile 0 (this wave's share) and the windows have landed
  __syncthreads();

  const int lane8 = lane * 8;
  for (int64_t t = 0; t < ntiles; ++t) {
    if (t + 1 < ntiles && DBG != 4) issue_tile(t + 1);
    const int kb0 = (int)(t * GI_KB);
    const int kb_end = kb0 + GI_KB;
    const int ldsB = (int)(lds0 + (unsigned)(t & 1) * GI_TILE);
    bool more = true;
    while (more) {
      // ---- phase A: my row's elements inside this tile (up to 4 per round) -------------------------------
      const int idx = (int)(p - wbase);
      int avail = GI_WIN - idx;
      if ((pend - p) < avail) avail = (int)(pend - p);
      int c[4];
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        c[i] = GI_SENT;
        v[i] = 0.f;
        if (i < avail) {
          c[i] = wcol[lane * GI_WSTRIDE + idx + i];
          v[i] = wval[lane * GI_WSTRIDE + idx + i];
        }
      }
      int k = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) k += (c[i] < kb_end) ? 1 : 0;
      const bool hit_window_end = (idx + k == GI_WIN) && (p + k < pend);
      const bool maybe_more = (k == 4) && !hit_window_end;
      int incl = k;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int n = __shfl_up(incl, off, 64);
        if (lane >= off) incl += n;
      }
      const int excl = incl - k;
      const int T = __builtin_amdgcn_readlane(incl, 63);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < k) {
          lrow[excl + i] = lane;
          loff[excl + i] = ldsB + (c[i] - kb0) * 512;
          lval[excl + i] = v[i];
        }
      }
      p += k;
      // ---- phase B: consume the T listed elements, 64 at a time ---------------------------------------------
      for (int base = 0; base < T; base += 64) {
        const int cnt = (T - base) < 64 ? (T - base) : 64;
        const int er = lrow[base + lane];
        const int eo = loff[base + lane];
        const float ev = lval[base + lane];
        if constexpr (ASM) {
          if (DBG != 3) gi_consume_asm(er, eo, __builtin_bit_cast(int, ev), cnt, lane8);
        } else {
          for (int e = 0; e < cnt; ++e) {
            const int r = __builtin_amdgcn_readlane(er, e);
            const int o = __builtin_amdgcn_readlane(eo, e);
            const float val = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ev), e));
            const float2 bb = *reinterpret_cast<const float2*>(lds + (o - (int)lds0) + lane8);
            accs[2 * r] = __builtin_fmaf(val, bb.x, accs[2 * r]);
            accs[2 * r + 1] = __builtin_fmaf(val, bb.y, accs[2 * r + 1]);
          }
        }
      }
      const bool any_refill = __any(hit_window_end);
      more = any_refill || __any(maybe_more);
      if (any_refill && DBG != 5) refill();
      if (DBG == 5) more = __any(maybe_more);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tile t+1 (this wave's share) has landed
    __syncthreads();
  }

  // ---- write the 64 rows of this wave ----------------------------------------------------------------------------
  int nvalid = (int)((M - row0) < GI_RW ? (M - row0) : (int64_t)GI_RW);
  if (nvalid < 0) nvalid = 0;
  if constexpr (ASM) {
    float* const obase = out + row0 * ldo + lane * 2;
    const int64_t stride_bytes = ldo * 4;
    asm volatile(
        "s_mov_b32 s36, 0\n\t"
        "v_mov_b32 v96, %0\n\t"
        "v_mov_b32 v97, %1\n\t"
        ".set spamd_gj, 128\n\t"
        ".rept 64\n\t"
        "s_cmp_ge_i32 s36, %3\n\t"
        "s_cbranch_scc1 9f\n\t"
        "global_store_dwordx2 v[96:97], v[spamd_gj:spamd_gj+1], off nt\n\t"
        "v_lshl_add_u64 v[96:97], %2, 0, v[96:97]\n\t"
        "s_add_i32 s36, s36, 1\n\t"
        ".set spamd_gj, spamd_gj+2\n\t"
        ".endr\n\t"
        "9:\n\t"
        :
        : "v"((unsigned)((uintptr_t)obase & 0xffffffffu)), "v"((unsigned)((uintptr_t)obase >> 32)), "s"(stride_bytes),
          "s"(nvalid)
        : "memory", "scc", "s36", "v96", "v97", GI_ACC_CLOBBERS);
  } else {
    for (int r = 0; r < nvalid; ++r) {
      float o[2] = {accs[2 * r], accs[2 * r + 1]};
      nt_store<float, 2>(out + (row0 + r) * ldo + lane * 2, o);
    }
  }
}

template <typename I, bool ASM, int DBG>
static int launch_gidx(int64_t M, int64_t K, const float* a_data, const I* a_idx, const I* a_ptr, const float* b,
                       int64_t ldb, float* out, int64_t ldo, hipStream_t s) {
  auto kern = &spmm_csr_gidx_kernel<I, ASM, DBG>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     GI_LDS);
  if (e != hipSuccess) return (int)e;
  const int64_t blocks = ceil_div(M, (int64_t)GI_RW * GI_WAVES);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), GI_LDS, s, M, K, a_data, a_idx, a_ptr, b, ldb, out,
                     ldo);
  return launch_status();
}

// fp32, N == 128, FMA mode, K < 2^22 (LDS offsets), 16-byte aligned B rows.  mode 1 = asm consume,
// mode 2 = debug consume (scratch accumulators; validates phase A / windows / tiles).
template <typename I>
int spmm_csr_gidx_dispatch(int64_t M, int64_t K, int64_t N, const float* a_data, const I* a_idx, const I* a_ptr,
                           const float* b, int64_t ldb, float* out, int64_t ldo, int mode, hipStream_t s) {
  if (N != 128 || K >= ((int64_t)1 << 31) || ((uintptr_t)b % 16) || (ldb % 4) || ((uintptr_t)out % 8) || (ldo % 2))
    return SPAMD_ETYPE;
  if (mode == 2) return launch_gidx<I, false, 0>(M, K, a_data, a_idx, a_ptr, b, ldb, out, ldo, s);
  if (mode == 3) return launch_gidx<I, true, 3>(M, K, a_data, a_idx, a_ptr, b, ldb, out, ldo, s);  // timing: no consume
  if (mode == 4) return launch_gidx<I, true, 4>(M, K, a_data, a_idx, a_ptr, b, ldb, out, ldo, s);  // timing: no tile DMA
  if (mode == 5) return launch_gidx<I, true, 5>(M, K, a_data, a_idx, a_ptr, b, ldb, out, ldo, s);  // timing: no refills
  return launch_gidx<I, true, 0>(M, K, a_data, a_idx, a_ptr, b, ldb, out, ldo, s);
}

template int spmm_csr_gidx_dispatch<int32_t>(int64_t, int64_t, int64_t, const float*, const int32_t*, const int32_t*,
                                             const float*, int64_t, float*, int64_t, int, hipStream_t);
template int spmm_csr_gidx_dispatch<int64_t>(int64_t, int64_t, int64_t, const float*, const int64_t*, const int64_t*,
                                             const float*, int64_t, float*, int64_t, int, hipStream_t);

}  // namespace spamd
