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
#include <mutex>
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

// ---- "row-vector" kernel: results at most 4 columns wide (N = 1 is the matrix-vector product) ----------------------
// With so few columns the row-group mapping above leaves 15 of 16 lanes idle.  Here the lanes run ALONG a row instead:
// L lanes (a power of two chosen per matrix, inside the kernel, from nnz / M = indptr[M] / M) share one compressed row;
// each lane takes the stored elements p, p + L, p + 2L, ... of it with coalesced loads of (index, value), gathers the NV
// entries of B's row and accumulates privately; a butterfly over the L lanes then forms the row's sums.  The products
// of one row are therefore added in a fixed TREE order, not k-ascending: results are deterministic and - for floating
// point - within rounding of the reference's (integers wrap identically), which is why the exact mode keeps the
// row-group kernel.  The pass is the CSR triplet's stream (12 or 8 B per stored element), once.
template <typename T, int NV>
__device__ __forceinline__ void rv_load(const T* p, bool aligned, T (&o)[NV]) {
  constexpr int BYTES = (int)sizeof(T) * NV;
  if constexpr (NV == 1) {
    o[0] = p[0];
  } else if constexpr (BYTES == 8 || BYTES == 16) {
    if (aligned) {
      const Vec<T, NV> v = *reinterpret_cast<const Vec<T, NV>*>(p);
#pragma unroll
      for (int e = 0; e < NV; ++e) o[e] = v.v[e];
      return;
    }
#pragma unroll
    for (int e = 0; e < NV; ++e) o[e] = p[e];
  } else if constexpr (BYTES == 32) {
    if (aligned) {
      const Vec<T, 2> v0 = *reinterpret_cast<const Vec<T, 2>*>(p);
      const Vec<T, 2> v1 = *reinterpret_cast<const Vec<T, 2>*>(p + 2);
      o[0] = v0.v[0]; o[1] = v0.v[1]; o[2] = v1.v[0]; o[3] = v1.v[1];
      return;
    }
#pragma unroll
    for (int e = 0; e < NV; ++e) o[e] = p[e];
  } else {
#pragma unroll
    for (int e = 0; e < NV; ++e) o[e] = p[e];
  }
}

template <typename T, typename I, int NV, int L>
__device__ __forceinline__ void rowvec_body(int64_t M, const T* __restrict__ a_data, const I* __restrict__ a_idx,
                                            const I* __restrict__ a_ptr, const T* __restrict__ b, int64_t ldb,
                                            T* __restrict__ out, int64_t ldo, bool aligned) {
  constexpr int RPW = SPAMD_WAVE / L;
  const int lane = threadIdx.x & (SPAMD_WAVE - 1);
  const int gl = lane & (L - 1);
  const int sub = lane / L;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x / SPAMD_WAVE) + (threadIdx.x / SPAMD_WAVE);
  const int64_t stride = (int64_t)gridDim.x * (blockDim.x / SPAMD_WAVE) * RPW;
  // A wave's chain  row pointers -> (index, value) -> gather -> butterfly  is three memory latencies long whatever the
  // row holds, and the CU's vector-memory queue is in order, so the chain is software-pipelined over the wave's row
  // slots: while slot t is being summed, the elements of slot t + 1 and the pointers of slot t + 2 are in flight.
  struct Elems {
    I i0, i1;
    T v0, v1;
  };
  auto ptrs = [&](int64_t rbase, int64_t& s, int64_t& e) {
    const int64_t r = rbase + sub;
    s = 0;
    e = 0;
    if (r < M) {
      s = (int64_t)a_ptr[r];
      e = (int64_t)a_ptr[r + 1];
    }
  };
  auto elems = [&](int64_t s, int64_t e, Elems& x) {
    const int64_t p = s + gl;
    x.i0 = 0; x.i1 = 0; x.v0 = T(0); x.v1 = T(0);
    if (p < e) {
      x.i0 = a_idx[p];
      x.v0 = a_data[p];
    }
    if (p + L < e) {
      x.i1 = a_idx[p + L];
      x.v1 = a_data[p + L];
    }
  };
  int64_t base = wave * RPW;
  int64_t s0, e0, s1, e1, s2, e2;
  Elems c, n;
  ptrs(base, s0, e0);
  ptrs(base + stride, s1, e1);
  elems(s0, e0, c);
  for (; base < M; base += stride) {
    ptrs(base + 2 * stride, s2, e2);
    elems(s1, e1, n);
    T acc[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) acc[e] = T(0);
    {
      const int64_t p = s0 + gl;
      T g0[NV], g1[NV];
      if (p < e0) rv_load<T, NV>(b + (int64_t)c.i0 * ldb, aligned, g0);
      if (p + L < e0) rv_load<T, NV>(b + (int64_t)c.i1 * ldb, aligned, g1);
      if (p < e0) {
#pragma unroll
        for (int e = 0; e < NV; ++e) acc[e] = mul_add<false>(c.v0, g0[e], acc[e]);
      }
      if (p + L < e0) {
#pragma unroll
        for (int e = 0; e < NV; ++e) acc[e] = mul_add<false>(c.v1, g1[e], acc[e]);
      }
    }
    // rows longer than 2 L (L is sized so that an average row is not)
    for (int64_t p = s0 + gl + 2 * L; p < e0; p += 2 * L) {
      const bool two = p + L < e0;
      const I j0 = a_idx[p];
      const T w0 = a_data[p];
      I j1 = 0;
      T w1 = T(0);
      if (two) {
        j1 = a_idx[p + L];
        w1 = a_data[p + L];
      }
      T r0[NV], r1[NV];
      rv_load<T, NV>(b + (int64_t)j0 * ldb, aligned, r0);
      if (two) rv_load<T, NV>(b + (int64_t)j1 * ldb, aligned, r1);
#pragma unroll
      for (int e = 0; e < NV; ++e) acc[e] = mul_add<false>(w0, r0[e], acc[e]);
      if (two) {
#pragma unroll
        for (int e = 0; e < NV; ++e) acc[e] = mul_add<false>(w1, r1[e], acc[e]);
      }
    }
    // every lane of the wave is back here: butterfly over the row's L lanes
#pragma unroll
    for (int off = L / 2; off >= 1; off >>= 1) {
#pragma unroll
      for (int e = 0; e < NV; ++e) acc[e] = (T)(acc[e] + lane_shfl(acc[e], lane ^ off));
    }
    if (base + sub < M && gl == 0) {
#pragma unroll
      for (int e = 0; e < NV; ++e) out[(base + sub) * ldo + e] = acc[e];
    }
    s0 = s1; e0 = e1; s1 = s2; e1 = e2;
    c = n;
  }
}

// lanes per row: the power of two with 2 L >= nnz / M (one trip covers an average row), 4..64
#define SPAMD_ROWVEC_PICK(B, LDB, AL)                                                                      \
  const int64_t avg = uniform((int64_t)a_ptr[M]) / M;                                                      \
  if (avg > 64) rowvec_body<T, I, NV, 64>(M, a_data, a_idx, a_ptr, B, LDB, out, ldo, AL);                  \
  else if (avg > 32) rowvec_body<T, I, NV, 32>(M, a_data, a_idx, a_ptr, B, LDB, out, ldo, AL);             \
  else if (avg > 16) rowvec_body<T, I, NV, 16>(M, a_data, a_idx, a_ptr, B, LDB, out, ldo, AL);             \
  else if (avg > 8) rowvec_body<T, I, NV, 8>(M, a_data, a_idx, a_ptr, B, LDB, out, ldo, AL);               \
  else rowvec_body<T, I, NV, 4>(M, a_data, a_idx, a_ptr, B, LDB, out, ldo, AL);

template <typename T, typename I, int NV>
__global__ void __launch_bounds__(256)
spmm_csr_rowvec_kernel(int64_t M, const T* __restrict__ a_data, const I* __restrict__ a_idx,
                       const I* __restrict__ a_ptr, const T* __restrict__ b, int64_t ldb, T* __restrict__ out,
                       int64_t ldo, int aligned) {
  SPAMD_ROWVEC_PICK(b, ldb, aligned != 0)
}

// The same with B resident in LDS (K * NV values, at most ROWVEC_LDS_BYTES): gathers of 4..32 bytes at random rows cost the
// CU's vector-memory pipe one cache line per lane (measured: 0.50 ms with them, 0.30 ms with coalesced stand-ins, config 2's
// matrix times a vector), the LDS serves them at bank rate.  Persistent blocks, as many per CU as the size of B allows
// (512 threads, up to four; one of 1024 threads above 80 KB); each copies B once.
constexpr int ROWVEC_LDS_BYTES = 160 * 1024;   // all of the CU's LDS (round 4; 144 KB before: config 2 x 4 fp32 columns is 156.25 KB)

template <typename T, typename I, int NV>
__global__ void __launch_bounds__(1024)
spmm_csr_rowvec_lds_kernel(int64_t M, int64_t K, const T* __restrict__ a_data, const I* __restrict__ a_idx,
                           const I* __restrict__ a_ptr, const T* __restrict__ b, int64_t ldb, T* __restrict__ out,
                           int64_t ldo) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rv_smem[];
  T* bl = reinterpret_cast<T*>(rv_smem);
  for (int64_t i = threadIdx.x; i < K * NV; i += blockDim.x) bl[i] = b[(i / NV) * ldb + (i % NV)];
  __syncthreads();
  SPAMD_ROWVEC_PICK(bl, (int64_t)NV, true)
}
#undef SPAMD_ROWVEC_PICK

constexpr int ROWVEC_MAX_N = 4;
#ifndef SPAMD_ROWVEC_LDS_MIN_M
#define SPAMD_ROWVEC_LDS_MIN_M 32768   // fewer rows: the blocks' copies of B cost more than the gathers they save
#endif

template <typename T, typename I>
static int launch_rowvec(int64_t M, int64_t K, int64_t N, const T* a_data, const I* a_idx, const I* a_ptr, const T* b,
                         int64_t ldb, T* out, int64_t ldo, hipStream_t s) {
  // (the lanes-per-row choice is the kernel's, so the grid is sized for the widest one - one row per wave - and
  // strides over the rows)
  const size_t rowbytes = sizeof(T) * (size_t)N;
  const size_t ldsbytes = rowbytes * (size_t)K;
  const bool lds = ldsbytes <= (size_t)ROWVEC_LDS_BYTES && M >= SPAMD_ROWVEC_LDS_MIN_M;
  const bool big = ldsbytes > 80 * 1024;
  int64_t blocks = ceil_div(M, 4);
  if (blocks > 256 * 8) blocks = 256 * 8;
  const size_t al = rowbytes > 16 ? 16 : rowbytes;
  const int aligned = (rowbytes == 8 || rowbytes == 16 || rowbytes == 32) && (uintptr_t)b % al == 0 && (ldb * sizeof(T)) % al == 0;
#define SPAMD_RV(NV)                                                                                              \
  case NV:                                                                                                        \
    if (lds) {                                                                                                    \
      auto kern = spmm_csr_rowvec_lds_kernel<T, I, NV>;                                                           \
      if (set_max_dynamic_lds((const void*)kern, ROWVEC_LDS_BYTES)) return SPAMD_EINVAL;                         \
      hipLaunchKernelGGL(kern, dim3(big ? 256 : 1024), dim3(big ? 1024 : 512), ldsbytes, s, M, K, a_data, a_idx, a_ptr, b, ldb, out, ldo); \
    } else {                                                                                                      \
      hipLaunchKernelGGL((spmm_csr_rowvec_kernel<T, I, NV>), dim3((unsigned)blocks), dim3(256), 0, s, M, a_data,  \
                         a_idx, a_ptr, b, ldb, out, ldo, aligned);                                                \
    }                                                                                                             \
    break;
  switch (N) {
    SPAMD_RV(1)
    SPAMD_RV(2)
    SPAMD_RV(3)
    SPAMD_RV(4)
    default: return SPAMD_EINVAL;
  }
#undef SPAMD_RV
  return launch_status();
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

extern "C" int spamd_spmm_csr_ldsb_fits(int val_dtype, int64_t M, int64_t K, int64_t N, const void* b, int64_t ldb,
                                        const void* out, int64_t ldo);
extern "C" int spamd_spmm_csr_ldsb(int val_dtype, int idx_dtype, int64_t M, int64_t K, int64_t N, const void* a_data,
                                   const void* a_indices, const void* a_indptr, const void* b, int64_t ldb, void* out,
                                   int64_t ldo, unsigned flags, void* stream);

extern "C" int spamd_spmm_csr_stream_fits(int val_dtype, int64_t M, int64_t K, int64_t N, const void* a_data,
                                          const void* a_indices);
extern "C" int spamd_spmm_csr_stream(int val_dtype, int idx_dtype, int64_t M, int64_t K, int64_t N, const void* a_data,
                                     const void* a_indices, const void* a_indptr, const void* b, int64_t ldb, void* out,
                                     int64_t ldo, int64_t nnz, unsigned flags, void* stream);

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
  // a short contracted axis and a result of at least half a 256-byte panel: B resident in LDS (spmm_ldsb.hip; same
  // k-ascending sums, so both arithmetic modes)
  if (!(flags & SPAMD_SPMM_ROWGROUP) && M >= 8192 && N * ((val_dtype == SPAMD_F64 || val_dtype == SPAMD_I64) ? 8 : 4) >= 128 &&
      spamd_spmm_csr_ldsb_fits(val_dtype, M, K, N, b, ldb, out, ldo))
    return spamd_spmm_csr_ldsb(val_dtype, idx_dtype, M, K, N, a_data, a_indices, a_indptr, b, ldb, out, ldo, flags, stream);
  // results of at most 4 columns, B fits LDS: the stream form (spmm_stream.hip; tree order per row like the row-vector kernel);
  // round 6: in several passes over chunks of columns too, while its passes over A cost less than the other kernels - up to
  // SPAMD_STREAM_MULTI_MAX_N columns in at most 3 passes (against the tiled executor's padded panel and the row-group kernel's
  // gathers), at most 4 columns of 8-byte values in 2 (against the row-vector kernel: 0.50 against 0.85-0.96 ms at config 2's
  // matrix; with 4-byte values two passes only tie with it)
  if (N <= SPAMD_STREAM_MULTI_MAX_N && !(flags & (SPAMD_SPMM_ROWGROUP | SPAMD_SPMM_ROWVEC)) && M >= SPAMD_ROWVEC_LDS_MIN_M && K > 0 &&
      !(exact && (val_dtype == SPAMD_F32 || val_dtype == SPAMD_F64))) {
    const int passes = spamd_spmm_csr_stream_fits(val_dtype, M, K, N, a_data, a_indices);
    const bool wide = val_dtype == SPAMD_F64 || val_dtype == SPAMD_I64;
    const int worth = N > ROWVEC_MAX_N ? 3 : (wide ? 2 : 1);
    if (passes >= 1 && passes <= worth)
      return spamd_spmm_csr_stream(val_dtype, idx_dtype, M, K, N, a_data, a_indices, a_indptr, b, ldb, out, ldo, -1, 0u, stream);
  }
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
      if (N <= ROWVEC_MAX_N && !(flags & SPAMD_SPMM_ROWGROUP)) return launch_rowvec<T, I>(M, K, N, ad, ai, ap, bb, ldb, oo, ldo, s);
      return dispatch_shape<T, I, false>(M, K, N, ad, ai, ap, bb, ldb, oo, ldo, s);
    })
  })
  return SPAMD_ETYPE;
}
