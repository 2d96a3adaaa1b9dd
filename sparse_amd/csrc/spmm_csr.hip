// A1: CSR x dense -> dense SpMM for gfx950 (replaces `_dot_csr_ndarray`,
// reference sparse/numba_backend/_common.py:720-755).
//
// Mapping ("row-group" kernel): G lanes of a 64-lane wave own one compressed row; each
// lane owns VEC contiguous output columns per column pass, so a wave-load of a B row is
// G*VEC*sizeof(T) contiguous bytes (512 B for fp32 N=128 with G=64, VEC=2).  The row's
// (index, value) pairs are fetched G at a time with one coalesced load and broadcast
// lane-by-lane (v_readlane for G=64, ds_bpermute otherwise).  Every output element is
// accumulated by exactly one lane in storage (k-ascending) order -> the summation order is
// the reference's, and the result is deterministic.  `out` is written exactly once per
// element (no zero-fill + read-modify-write as in the reference).
#include "common.h"
#include <stdlib.h>
#include <string.h>

namespace spamd {

template <typename T, typename I, int VEC, int G, bool EXACT, int UNROLL, int CH>
__global__ void __launch_bounds__(256)
spmm_csr_rowgroup_kernel(int64_t M, int64_t N, const T* __restrict__ a_data,
                         const I* __restrict__ a_idx, const I* __restrict__ a_ptr,
                         const T* __restrict__ b, int64_t ldb, T* __restrict__ out,
                         int64_t ldo, int64_t panel, int64_t k_lo, int64_t k_hi, int accumulate) {
  // Only stored elements with column index in [k_lo, k_hi) are applied (K-split passes keep the
  // gathered part of B inside the 4 MiB per-XCD L2); with `accumulate` the pass continues from
  // the partial sums already in `out` — same k-ascending order, so results are unchanged.
  // blockIdx.y selects a column panel [c_lo, c_hi) of the output.  Inside a panel a lane owns
  // CH groups of VEC contiguous columns, G*VEC columns apart, so one pass over the row's stored
  // elements covers G*VEC*CH columns (wide outputs such as the 512-column tensordot config do
  // not re-read A per 128 columns).
  const int64_t c_lo = (int64_t)blockIdx.y * panel;
  const int64_t c_hi = (c_lo + panel < N) ? (c_lo + panel) : N;
  constexpr int RPW = SPAMD_WAVE / G;  // rows per wave
  using V = Vec<T, VEC>;
  const int lane = threadIdx.x & (SPAMD_WAVE - 1);
  const int gl = lane % G;             // lane within the row group
  const int gbase = lane - gl;         // first lane of my group
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x / SPAMD_WAVE) + (threadIdx.x / SPAMD_WAVE);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x / SPAMD_WAVE);

  for (int64_t row0 = wave * RPW; row0 < M; row0 += nwaves * RPW) {
    const int64_t row = row0 + lane / G;
    const bool row_ok = row < M;
    const int64_t start = row_ok ? (int64_t)a_ptr[row] : 0;
    const int64_t end = row_ok ? (int64_t)a_ptr[row + 1] : 0;

    for (int64_t c0 = c_lo; c0 < c_hi; c0 += (int64_t)G * VEC * CH) {
      int64_t col[CH];
      bool col_ok[CH];  // N % VEC == 0 and panel % VEC == 0 (dispatcher)
      T acc[CH][VEC];
#pragma unroll
      for (int h = 0; h < CH; ++h) {
        col[h] = c0 + (int64_t)h * G * VEC + (int64_t)gl * VEC;
        col_ok[h] = col[h] < c_hi;
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[h][e] = T(0);
        if (accumulate && row_ok && col_ok[h]) {
          const V o = *reinterpret_cast<const V*>(out + row * ldo + col[h]);
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[h][e] = o.v[e];
        }
      }

      for (int64_t p = start; p < end; p += G) {
        const int64_t mine = p + gl;
        I ci = 0;
        T vi = T(0);
        if (mine < end) {
          ci = a_idx[mine];
          vi = a_data[mine];
        }
        int cnt = (int)((end - p) < (int64_t)G ? (end - p) : (int64_t)G);
        constexpr int U = (UNROLL / CH) < 1 ? 1 : (UNROLL / CH);  // keep ~UNROLL gathers in flight
        int j = 0;
        if (k_hi >= 0) {
          // sorted columns: the elements of this chunk inside [k_lo, k_hi) are one contiguous run
          const bool in = (mine < end) && ((int64_t)ci >= k_lo) && ((int64_t)ci < k_hi);
          const unsigned long long bal = __ballot(in);
          const unsigned long long grp = (G == SPAMD_WAVE) ? bal : ((bal >> gbase) & ((1ull << (G & 63)) - 1ull));
          j = grp ? __builtin_ctzll(grp) : 0;
          cnt = j + __builtin_popcountll(grp);
        }
        for (; j + U <= cnt; j += U) {
          I cj[U];
          T vj[U];
          V bj[U][CH];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if constexpr (G == SPAMD_WAVE) {
              cj[u] = wave_bcast(ci, j + u);
              vj[u] = wave_bcast(vi, j + u);
            } else {
              cj[u] = lane_shfl(ci, gbase + j + u);
              vj[u] = lane_shfl(vi, gbase + j + u);
            }
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int h = 0; h < CH; ++h)
              if (col_ok[h]) bj[u][h] = *reinterpret_cast<const V*>(b + (int64_t)cj[u] * ldb + col[h]);
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int h = 0; h < CH; ++h) {
              if (col_ok[h]) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[h][e] = mul_add<EXACT>(vj[u], bj[u][h].v[e], acc[h][e]);
              }
            }
          }
        }
        for (; j < cnt; ++j) {
          I cj;
          T vj;
          if constexpr (G == SPAMD_WAVE) {
            cj = wave_bcast(ci, j);
            vj = wave_bcast(vi, j);
          } else {
            cj = lane_shfl(ci, gbase + j);
            vj = lane_shfl(vi, gbase + j);
          }
#pragma unroll
          for (int h = 0; h < CH; ++h) {
            if (col_ok[h]) {
              V bj = *reinterpret_cast<const V*>(b + (int64_t)cj * ldb + col[h]);
#pragma unroll
              for (int e = 0; e < VEC; ++e) acc[h][e] = mul_add<EXACT>(vj, bj.v[e], acc[h][e]);
            }
          }
        }
      }
#pragma unroll
      for (int h = 0; h < CH; ++h)
#ifdef SPMM_PLAIN_STORE
        if (row_ok && col_ok[h]) {
          V o;
#pragma unroll
          for (int e = 0; e < VEC; ++e) o.v[e] = acc[h][e];
          *reinterpret_cast<V*>(out + row * ldo + col[h]) = o;
        }
#else
        if (row_ok && col_ok[h]) nt_store<T, VEC>(out + row * ldo + col[h], acc[h]);
#endif
    }
  }
}

struct SpmmVariant {
  int g = 0, vec = 0, unroll = 0;
  int64_t panel = 0;  // 0 = whole N in one pass
  int ch = 0, ksplit = 0;
};

// Tuning hook, compiled ONLY into -DSPAMD_TUNING builds (tools/build_variant.sh): SPAMD_SPMM_VARIANT="G=32,VEC=2,U=8,PANEL=64"
// overrides the heuristic.  The shipped library never reads the environment.
static SpmmVariant env_variant() {
  SpmmVariant v;
#ifdef SPAMD_TUNING
  const char* e = getenv("SPAMD_SPMM_VARIANT");
  if (!e) return v;
  const char* p;
  if ((p = strstr(e, "G="))) v.g = atoi(p + 2);
  if ((p = strstr(e, "VEC="))) v.vec = atoi(p + 4);
  if ((p = strstr(e, "U="))) v.unroll = atoi(p + 2);
  if ((p = strstr(e, "PANEL="))) v.panel = atoll(p + 6);
  if ((p = strstr(e, "CH="))) v.ch = atoi(p + 3);
  if ((p = strstr(e, "KSPLIT="))) v.ksplit = atoi(p + 7);
#endif
  return v;
}

template <typename T, typename I, int VEC, int G, bool EXACT, int UNROLL, int CH>
static int launch_rowgroup(int64_t M, int64_t N, const T* a_data, const I* a_idx, const I* a_ptr,
                           const T* b, int64_t ldb, T* out, int64_t ldo, int64_t panel,
                           int g_ksplit, int64_t g_K, hipStream_t s) {
  constexpr int RPW = SPAMD_WAVE / G;
  constexpr int WPB = 4;  // waves per 256-thread block
  int64_t blocks = ceil_div(M, (int64_t)RPW * WPB);
  const int64_t cap = 256 * 8 * 4;  // grid-stride above this many blocks
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  if (panel <= 0 || panel > N) panel = N;
  const unsigned npanels = (unsigned)ceil_div(N, panel);
  const int ks = g_ksplit < 1 ? 1 : g_ksplit;
  for (int pass = 0; pass < ks; ++pass) {
    const int64_t k_lo = ks == 1 ? 0 : (g_K * pass) / ks;
    const int64_t k_hi = ks == 1 ? -1 : (pass == ks - 1 ? (int64_t)1 << 62 : (g_K * (pass + 1)) / ks);
    hipLaunchKernelGGL((spmm_csr_rowgroup_kernel<T, I, VEC, G, EXACT, UNROLL, CH>),
                       dim3((unsigned)blocks, npanels), dim3(256), 0, s, M, N, a_data, a_idx, a_ptr, b,
                       ldb, out, ldo, panel, k_lo, k_hi, pass > 0 ? 1 : 0);
    if (int rc = launch_status()) return rc;
  }
  return 0;
}

template <typename T, typename I, bool EXACT>
static int dispatch_shape(int64_t M, int64_t K, int64_t N, const T* a_data, const I* a_idx, const I* a_ptr,
                          const T* b, int64_t ldb, T* out, int64_t ldo, hipStream_t s) {
  // widest vector (<= 16 B) that the shapes/alignments allow
  int vmax = 16 / (int)sizeof(T);
  if (vmax > 4) vmax = 4;
  auto ok = [&](int v) {
    return N % v == 0 && ldb % v == 0 && ldo % v == 0 &&
           ((uintptr_t)b % (v * sizeof(T))) == 0 && ((uintptr_t)out % (v * sizeof(T))) == 0;
  };
  while (vmax > 1 && !ok(vmax)) vmax >>= 1;
  // Measured on MI355X (tools/micro/gather_bw.hip, profiles/): the vector-memory front end
  // moves 16 B per lane per instruction at ~25 TB/s chip-wide but 8 B per lane at only
  // ~18.5 TB/s, so every lane always fetches the widest vector the alignment allows and a row
  // takes just as many lanes as that needs — 32 lanes for 128 fp32 columns, i.e. two rows per
  // wave.  The group size is the power of two (16/32/64) that covers N in one pass if possible.
  int vec = vmax, g = 64, unroll = 8;
  int64_t panel = N;
  {
    const int64_t lanes = ceil_div(N, vec);
    g = lanes <= 16 ? 16 : (lanes <= 32 ? 32 : 64);
  }
  const SpmmVariant ev = env_variant();
  if (ev.g) g = ev.g;
  if (ev.vec && ev.vec <= vmax) vec = ev.vec;
  if (ev.unroll) unroll = ev.unroll;
  if (ev.panel > 0 && ev.panel % vec == 0) panel = ev.panel;
  // column groups per lane: cover the panel in one pass over A when it is at most 4 groups wide
  int ch = 1;
  {
    const int64_t groups = ceil_div(panel, (int64_t)g * vec);
    ch = groups >= 4 ? 4 : (groups >= 2 ? 2 : 1);
  }
  if (ev.ch) ch = ev.ch;
  // K-split passes (tuning hook only): measured slower than one pass on MI355X (2 passes 3.1 ms
  // vs 2.6 ms at config 2) — the kernel is bound by per-row issue overheads, not by L2 misses.
  const int ksplit = ev.ksplit > 0 ? ev.ksplit : 1;
#define SPAMD_CASE(V, GG)                                                                        \
  if (vec == V && g == GG) {                                                                     \
    if (ch == 4)                                                                                 \
      return launch_rowgroup<T, I, V, GG, EXACT, 8, 4>(M, N, a_data, a_idx, a_ptr, b, ldb, out,  \
                                                       ldo, panel, ksplit, K, s);                           \
    if (ch == 2)                                                                                 \
      return launch_rowgroup<T, I, V, GG, EXACT, 8, 2>(M, N, a_data, a_idx, a_ptr, b, ldb, out,  \
                                                       ldo, panel, ksplit, K, s);                           \
    if (unroll == 4)                                                                             \
      return launch_rowgroup<T, I, V, GG, EXACT, 4, 1>(M, N, a_data, a_idx, a_ptr, b, ldb, out,  \
                                                       ldo, panel, ksplit, K, s);                           \
    return launch_rowgroup<T, I, V, GG, EXACT, 8, 1>(M, N, a_data, a_idx, a_ptr, b, ldb, out,    \
                                                     ldo, panel, ksplit, K, s);                             \
  }
  SPAMD_CASE(1, 64)
  SPAMD_CASE(2, 64)
  SPAMD_CASE(1, 32)
  SPAMD_CASE(2, 32)
  SPAMD_CASE(1, 16)
  SPAMD_CASE(2, 16)
  if constexpr (sizeof(T) == 4) {
    SPAMD_CASE(4, 64)
    SPAMD_CASE(4, 32)
    SPAMD_CASE(4, 16)
  }
#undef SPAMD_CASE
  return SPAMD_EINVAL;
}

}  // namespace spamd

extern "C" int spamd_spmm_csr(int val_dtype, int idx_dtype, int64_t M, int64_t K, int64_t N,
                              const void* a_data, const void* a_indices, const void* a_indptr,
                              const void* b, int64_t ldb, void* out, int64_t ldo, unsigned flags,
                              void* stream) {
  using namespace spamd;
  if (M < 0 || K < 0 || N < 0) return SPAMD_EINVAL;
  if (M == 0 || N == 0) return 0;
  if (!a_indptr || !out || ldo < N || (K > 0 && (!b || ldb < N))) return SPAMD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const bool exact = (flags & SPAMD_EXACT_MULADD) != 0;
  SPAMD_DISPATCH_VAL(val_dtype, T, {
    SPAMD_DISPATCH_IDX(idx_dtype, I, {
      const T* ad = (const T*)a_data;
      const I* ai = (const I*)a_indices;
      const I* ap = (const I*)a_indptr;
      const T* bb = (const T*)b;
      T* oo = (T*)out;
      if constexpr (std::is_floating_point<T>::value) {
        if (exact) return dispatch_shape<T, I, true>(M, K, N, ad, ai, ap, bb, ldb, oo, ldo, s);
      }
      return dispatch_shape<T, I, false>(M, K, N, ad, ai, ap, bb, ldb, oo, ldo, s);
    })
  })
  return SPAMD_ETYPE;
}
