// A9: sampled dense-dense matrix multiplication,  out[n] = s[n] * sum_k A[i_n, k] * Bt[j_n, k].
//
// The reference has no SDDMM kernel: `s * (a @ b)` (examples/sddmm_example.py:51-52) forms the
// full dense M x N product with BLAS and then gathers it at the mask's coordinates
// (`_Elemwise`, _umath.py:602-633) — 10^10 elements for BASELINE config 4.  Here only the
// sampled dot products are formed: LPN lanes of a wave own one stored element, stream the two
// K-long rows with 16-byte loads (B is taken K-major, i.e. as Bt = B^T row-major, so both rows
// are contiguous), multiply-accumulate in fp32 (bf16/fp32 inputs) or fp64, and reduce across
// the LPN lanes with wave shuffles.  Gather-bound (L2): 2*K*sizeof(in) bytes per element.
#include "sddmm_common.h"
#include <stdlib.h>
#include <algorithm>

namespace spamd {

// TIN: element type of A/Bt; TS: type of the mask values and of the output; LPN lanes per element;
// UNR stored elements per lane group in flight (all their row loads are issued before any FMA).
template <typename TIN, typename TS, typename I, int LPN, int UNR>
__global__ void __launch_bounds__(256)
sddmm_kernel(int64_t nnz, const I* __restrict__ rows, const I* __restrict__ cols, const TS* __restrict__ s_data,
             const TIN* __restrict__ A, int64_t lda, const TIN* __restrict__ Bt, int64_t ldb, int64_t K,
             TS* __restrict__ out) {
  using ACC = typename Acc<TIN>::type;
  constexpr int EPL = 16 / (int)sizeof(TIN);  // elements per 16-byte load
  using VT = Vec<TIN, EPL>;
  const int lane = threadIdx.x & 63;
  const int sub = lane % LPN;
  const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPN;
  const int64_t ngroups = (int64_t)gridDim.x * blockDim.x / LPN;
  const int64_t kvec = (K / EPL) * EPL;
  for (int64_t n0 = group * UNR; n0 < nnz; n0 += ngroups * UNR) {
    const TIN* ar[UNR];
    const TIN* br[UNR];
    ACC acc[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int64_t n = (n0 + u < nnz) ? (n0 + u) : (nnz - 1);  // clamp: duplicates are not stored
      ar[u] = A + (int64_t)rows[n] * lda;
      br[u] = Bt + (int64_t)cols[n] * ldb;
      acc[u] = 0;
    }
    for (int64_t k = (int64_t)sub * EPL; k + EPL <= K; k += (int64_t)LPN * EPL) {
      VT av[UNR], bv[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        av[u] = *reinterpret_cast<const VT*>(ar[u] + k);
        bv[u] = *reinterpret_cast<const VT*>(br[u] + k);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[u] = __builtin_fma(to_acc(av[u].v[e]), to_acc(bv[u].v[e]), acc[u]);
      }
    }
    // tail (K not a multiple of the vector width): scalar, spread over the group's lanes
#pragma unroll
    for (int u = 0; u < UNR; ++u)
      for (int64_t kk = kvec + sub; kk < K; kk += LPN) acc[u] = __builtin_fma(to_acc(ar[u][kk]), to_acc(br[u][kk]), acc[u]);
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
#pragma unroll
      for (int off = LPN / 2; off > 0; off >>= 1) acc[u] += __shfl_xor(acc[u], off, 64);
      if (sub == 0 && n0 + u < nnz) out[n0 + u] = (TS)((ACC)s_data[n0 + u] * acc[u]);
    }
  }
}

#ifndef SD_NT_A
#define SD_NT_A 0
#endif
#ifndef SD_NT_MASK
#define SD_NT_MASK 1
#endif

// STREAM_A: the A rows are read with the non-temporal hint (panel order: a row of A is used by ONE lane group once per
// panel and the panel's Bt rows are what must stay in L2; in row-major order A rows are left to the default policy).
template <typename TIN, typename I, int LPN, int KS, int U0, bool STREAM_A>
__device__ __forceinline__ void sd_batch4(int cnt, int sub, I myrow, I mycol, const char* Ab, const char* Bb,
                                          int64_t lda_b, int64_t ldb_b, int64_t koff_b, I& cur,
                                          Vec<TIN, 16 / (int)sizeof(TIN)> (&av)[KS], typename Acc<TIN>::type& res) {
  using ACC = typename Acc<TIN>::type;
  using VT = Vec<TIN, 16 / (int)sizeof(TIN)>;
  constexpr int64_t step_b = (int64_t)LPN * 16;
  VT bv[4][KS];
  I r[4];
#define SD_LOAD(k)                                                                                            \
  {                                                                                                           \
    const I c = sd_bcast<LPN, U0 + k>(mycol);                                                                 \
    r[k] = sd_bcast<LPN, U0 + k>(myrow);                                                                      \
    const char* bp = Bb + ((int64_t)c * ldb_b + koff_b);                                                      \
    _Pragma("unroll") for (int s = 0; s < KS; ++s) bv[k][s] = *reinterpret_cast<const VT*>(bp + s * step_b); \
  }
  SD_LOAD(0) SD_LOAD(1) SD_LOAD(2) SD_LOAD(3)
#undef SD_LOAD
#define SD_DOT(k)                                                                                             \
  if (U0 + k < cnt) {                                                                                         \
    if (r[k] != cur) {                                                                                        \
      cur = r[k];                                                                                             \
      const char* ap = Ab + ((int64_t)cur * lda_b + koff_b);                                                  \
      _Pragma("unroll") for (int s = 0; s < KS; ++s)                                                          \
        av[s] = STREAM_A ? sd_load_nt<VT>(ap + s * step_b) : *reinterpret_cast<const VT*>(ap + s * step_b);  \
    }                                                                                                         \
    const ACC t = sd_dot_group<TIN, VT, LPN, KS>(av, bv[k]);                                                  \
    res = sub == U0 + k ? t : res;                                                                            \
  }
  SD_DOT(0) SD_DOT(1) SD_DOT(2) SD_DOT(3)
#undef SD_DOT
}

template <typename TIN, typename I, int LPN, int KS, int U0, bool STREAM_A>
struct SdStep {
  template <typename... Args>
  static __device__ __forceinline__ void run(int cnt, Args&... args) {
    if constexpr (U0 < LPN) {
      if (U0 < cnt) sd_batch4<TIN, I, LPN, KS, U0, STREAM_A>(cnt, args...);
      SdStep<TIN, I, LPN, KS, U0 + 4, STREAM_A>::run(cnt, args...);
    }
  }
};

// Row-cached form for masks stored in row-major order (canonical COO / CSR order): a lane group walks a
// CONTIGUOUS chunk of stored elements LPN at a time.  Lane u of the group loads element u's row, column, mask value
// (and output position) - one coalesced load per array and step - and the group then takes the elements one after the
// other: row/column are broadcast across the group (DPP), the A row of consecutive elements is usually the same one
// and stays in registers (KS vectors per lane), so only the Bt rows are gathered (UNR of them in flight); the group's
// sum lands in lane u, and ONE store per step writes the LPN results.  K == LPN * KS * EPL exactly.  Any element
// order is correct; only row-major order is fast.
//
// PERM (column-panel order, see spamd_sddmm_panels): rows/cols/s_data are given in an order that walks the mask one
// panel of Bt rows at a time (row-major inside a panel), `perm[n]` is the element's position in out, and every lane
// group takes ONE short chunk: the workgroups resident at any moment - handed out in order - then cover one contiguous
// window of that order, so the panel's Bt rows (a few MB) stay in every XCD's L2 instead of being fetched from the
// Infinity Cache for each element.
template <typename TIN, typename TS, typename I, int LPN, int KS, int UNR, bool PERM>
__global__ void __launch_bounds__(256)
sddmm_rowcache_kernel(int64_t nnz, int64_t chunk, const I* __restrict__ rows, const I* __restrict__ cols,
                      const TS* __restrict__ s_data, const TIN* __restrict__ A, int64_t lda,
                      const TIN* __restrict__ Bt, int64_t ldb, TS* __restrict__ out, const int64_t* __restrict__ perm,
                      const int64_t* __restrict__ xstate) {
  using ACC = typename Acc<TIN>::type;
  constexpr int EPL = 16 / (int)sizeof(TIN);
  using VT = Vec<TIN, EPL>;
  static_assert(UNR == 4 && LPN % UNR == 0, "whole batches of four per step");
  const int sub = (threadIdx.x & 63) % LPN;
  // elements [cbeg0, cbeg0 + chunk), then every `cstride`-th chunk after it, below nnz_end
  int64_t cbeg0 = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPN) * chunk;
  int64_t cstride = ((int64_t)gridDim.x * blockDim.x / LPN) * chunk;
  int64_t nnz_end = nnz;
  if constexpr (PERM) {
    // XCD-private panels (xstate = first[9]): the element order is XCD-major — the elements of the panels that belong
    // to XCD x are [first[x], first[x + 1]) — and workgroup b takes piece b / 8 of the range of XCD b % 8, the XCD it is
    // observed to run on (MI355X_MICROARCH.md: for speed only, nothing depends on it), so that a panel's Bt rows are
    // fetched into ONE L2 instead of eight.  (Reading HW_REG_XCC_ID and taking a ticket per piece instead was built and
    // is slower, 0.51 vs 0.36 ms at config 4: every short-lived workgroup then starts with an atomic's round trip.)
    if (xstate) {
      __shared__ int64_t piece_s[2];
      const int64_t piece = (int64_t)(blockDim.x / LPN) * chunk;
      const int x = (int)(blockIdx.x & 7u);
      const int64_t lo = (int64_t)xstate[x], hi = (int64_t)xstate[x + 1];
      const int64_t b = lo + (int64_t)(blockIdx.x >> 3) * piece;
      if (threadIdx.x == 0) {
        piece_s[0] = b < hi ? b : 0;
        piece_s[1] = b < hi ? (b + piece < hi ? b + piece : hi) : 0;
      }
      __syncthreads();
      cbeg0 = piece_s[0] + (int64_t)(threadIdx.x / LPN) * chunk;
      cstride = (int64_t)1 << 40;   // one chunk per lane group
      nnz_end = piece_s[1];
    }
  }
  const char* const Ab = reinterpret_cast<const char*>(A);
  const char* const Bb = reinterpret_cast<const char*>(Bt);
  const int64_t lda_b = lda * (int64_t)sizeof(TIN), ldb_b = ldb * (int64_t)sizeof(TIN);
  const int64_t koff_b = (int64_t)sub * 16;
  I cur = (I)-1;
  VT av[KS];
  for (int64_t cbeg = cbeg0; cbeg < nnz_end; cbeg += cstride) {
    const int64_t cend = cbeg + chunk < nnz_end ? cbeg + chunk : nnz_end;
    for (int64_t nbeg = cbeg; nbeg < cend; nbeg += LPN) {
      const int cnt = (int)(cend - nbeg < LPN ? cend - nbeg : LPN);  // uniform inside the group
      const bool mine = sub < cnt;
      const int64_t nl = nbeg + (mine ? sub : 0);
      // (panel order: the mask's arrays are a once-through stream as well)
      constexpr bool NTM = PERM && SD_NT_MASK;
      const I myrow = NTM ? __builtin_nontemporal_load(rows + nl) : rows[nl];
      const I mycol = NTM ? __builtin_nontemporal_load(cols + nl) : cols[nl];
      const TS mys = NTM ? __builtin_nontemporal_load(s_data + nl) : s_data[nl];
      int64_t mypos = nl;
      if constexpr (PERM) mypos = NTM ? __builtin_nontemporal_load(perm + nl) : perm[nl];
      ACC res = 0;
      int lane_in_group = sub;
      SdStep<TIN, I, LPN, KS, 0, PERM && SD_NT_A>::run(cnt, lane_in_group, myrow, mycol, Ab, Bb, lda_b, ldb_b, koff_b, cur, av, res);
      if (mine) {
        const TS v = (TS)((ACC)mys * res);
        if constexpr (PERM) __builtin_nontemporal_store(v, out + mypos);  // scattered: keep these lines from displacing the Bt panel in L2
        else out[mypos] = v;
      }
    }
  }
}

template <typename TIN, typename TS, typename I>
static int launch_sddmm(int64_t nnz, const I* rows, const I* cols, const TS* s, const TIN* A, int64_t lda,
                        const TIN* Bt, int64_t ldb, int64_t K, TS* out, hipStream_t st, const int64_t* perm = nullptr,
                        int64_t perm_chunk = 0, const int64_t* xstate = nullptr, int64_t xmax = 0) {
  constexpr int EPL = 16 / (int)sizeof(TIN);
  const int64_t vecs = K / EPL;
  int lpn = 4;
  while (lpn < 64 && vecs > lpn * 2) lpn <<= 1;  // ~2 vector loads per lane per operand
  // (the gather kernel at the bottom — any K — measured on MI355X at config 4 in round 1: bf16 rows are fastest with one
  // element per lane group in flight, fp32/fp64 rows with four)
  {
    // row-cached kernel (config 4: 0.70 ms in the mask's own order, 0.36 ms in XCD-private panel order): K must be
    // LPN * KS vectors exactly (KS in 1, 2, 4); try 16 lanes per element first
#ifdef SPAMD_TUNING
    const char* v = getenv("SPAMD_SDDMM_VARIANT");  // tuning hook (-DSPAMD_TUNING builds only): "0" = gather kernel only
    const bool allow = !(v && v[0] == '0');
#else
    constexpr bool allow = true;
#endif
    for (int L = 16; allow && L <= 64; L <<= 1) {
      if (vecs % L) continue;
      const int ks = (int)(vecs / L);
      if (ks < 1 || ks > 4) continue;       // (3: rows of 768 / 1536 / 3072 bytes - K = 192, 384, 768 in fp32 - since round 6)
      constexpr int U = 4;
      const int64_t groups_wanted = 256 * 16 * (256 / L);  // 16 workgroups per CU
      int64_t chunk = ceil_div(nnz, groups_wanted);
      chunk = ceil_div(chunk, (int64_t)L) * L;  // whole steps of L elements
      if (perm) {
        const int64_t want = perm_chunk > 0 ? ceil_div(perm_chunk, (int64_t)L) * L : (int64_t)L;
        if (chunk > want) chunk = want;
      }
      // (column-panel order: one chunk per lane group, so that the workgroups resident at any moment - handed out in
      // order - cover one contiguous window of the element order, however unevenly they progress)
      const int64_t groups = perm ? ceil_div(nnz, chunk) : std::min(ceil_div(nnz, chunk), groups_wanted);
      int64_t blocks = ceil_div(groups * L, (int64_t)256);
      if (perm && xstate)   // a piece per workgroup, eight workgroups (one per XCD) per piece index
        blocks = 8 * std::max<int64_t>(ceil_div(xmax, (int64_t)(256 / L) * chunk), 1);
#define SDL(LL, KK, PP)                                                                                       \
  hipLaunchKernelGGL((sddmm_rowcache_kernel<TIN, TS, I, LL, KK, U, PP>), dim3((unsigned)blocks), dim3(256), 0, st, \
                     nnz, chunk, rows, cols, s, A, lda, Bt, ldb, out, perm, xstate)
#define SDR(LL, KK)                                                                                           \
  if (L == LL && ks == KK) {                                                                                  \
    if constexpr (LL * KK * 16 >= 1024) {   /* (panel order for shorter rows: sddmm_panel.hip) */            \
      if (perm) SDL(LL, KK, true);                                                                            \
      else SDL(LL, KK, false);                                                                                \
    } else {                                                                                                  \
      if (perm) return SPAMD_EINVAL;                                                                          \
      SDL(LL, KK, false);                                                                                     \
    }                                                                                                         \
    return launch_status();                                                                                   \
  }
      SDR(16, 1) SDR(16, 2) SDR(16, 3) SDR(16, 4) SDR(32, 1) SDR(32, 2) SDR(32, 3) SDR(32, 4) SDR(64, 1) SDR(64, 2) SDR(64, 3) SDR(64, 4)
#undef SDR
#undef SDL
    }
  }
  if (perm) return SPAMD_EINVAL;  // the panel order exists for the row-cached kernel only
  constexpr int UNR = sizeof(TIN) >= 4 ? 4 : 1;
  int64_t blocks = ceil_div(ceil_div(nnz, UNR) * lpn, 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
#define SD(L)                                                                                              \
  if (lpn == L) {                                                                                          \
    hipLaunchKernelGGL((sddmm_kernel<TIN, TS, I, L, UNR>), dim3((unsigned)blocks), dim3(256), 0, st, nnz, rows, \
                       cols, s, A, lda, Bt, ldb, K, out);                                                  \
    return launch_status();                                                                                \
  }
  SD(4) SD(8) SD(16) SD(32) SD(64)
#undef SD
  return SPAMD_EINVAL;
}

}  // namespace spamd

using namespace spamd;

template <typename I>
__global__ void sddmm_panel_keys_kernel(int64_t nnz, const I* __restrict__ cols, int64_t width, int64_t per_xcd,
                                        int64_t* __restrict__ keys) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n < nnz) {
    const int64_t p = (int64_t)cols[n] / width;
    keys[n] = per_xcd > 0 ? (p % 8) * per_xcd + p / 8 : p;
  }
}

// keys[n] = cols[n] / width: the column panel of each stored element (stable sort by it = panel order).  With
// per_xcd > 0 (= ceil(panels / 8)) the key is XCD-major instead: panel p belongs to XCD p % 8 and is that XCD's
// (p / 8)-th panel, key = (p % 8) * per_xcd + p / 8, so that the elements of one XCD's panels are contiguous.
extern "C" int spamd_sddmm_panel_keys(int idx_dtype, int64_t nnz, const void* cols, int64_t width, int64_t per_xcd,
                                      void* keys, void* stream) {
  if (nnz < 0 || width < 1 || per_xcd < 0) return SPAMD_EINVAL;
  if (nnz == 0) return 0;
  SPAMD_DISPATCH_IDX(idx_dtype, I, {
    hipLaunchKernelGGL((sddmm_panel_keys_kernel<I>), dim3((unsigned)ceil_div(nnz, (int64_t)256)), dim3(256), 0,
                       (hipStream_t)stream, nnz, (const I*)cols, width, per_xcd, (int64_t*)keys);
    return launch_status();
  })
  return SPAMD_ETYPE;
}

static int sddmm_entry(int in_dtype, int s_dtype, int idx_dtype, int64_t nnz, const void* rows, const void* cols,
                       const void* s_data, const void* A, int64_t lda, const void* Bt, int64_t ldb, int64_t K,
                       void* out, void* stream, const int64_t* perm, int64_t perm_chunk, const int64_t* xstate,
                       int64_t xmax);

extern "C" int spamd_sddmm(int in_dtype, int s_dtype, int idx_dtype, int64_t nnz, const void* rows, const void* cols,
                           const void* s_data, const void* A, int64_t lda, const void* Bt, int64_t ldb, int64_t K,
                           void* out, void* stream) {
  return sddmm_entry(in_dtype, s_dtype, idx_dtype, nnz, rows, cols, s_data, A, lda, Bt, ldb, K, out, stream, nullptr, 0,
                     nullptr, 0);
}

// 1 when K elements of in_dtype have a row-cached kernel (what spamd_sddmm_panels needs), else 0.
extern "C" int spamd_sddmm_has_panels(int in_dtype, int64_t K) {
  const int esz = in_dtype == SPAMD_BF16 ? 2 : (in_dtype == SPAMD_F32 ? 4 : (in_dtype == SPAMD_F64 ? 8 : 0));
  if (!esz || K <= 0 || (K * esz) % 16) return 0;
  const int64_t vecs = K * esz / 16;
  for (int L = 16; L <= 64; L <<= 1) {
    if (vecs % L) continue;
    const int64_t ks = vecs / L;
    if (ks >= 1 && ks <= 4) return 1;
  }
  return 0;
}

// the column-panel order for rows of 1 KB and more (called by spamd_sddmm_panels, sddmm_panel.hip)
int sddmm_rowcache_panels(int in_dtype, int s_dtype, int idx_dtype, int64_t nnz, const void* rows, const void* cols,
                          const void* s_data, const void* A, int64_t lda, const void* Bt, int64_t ldb, int64_t K, void* out,
                          void* stream, const int64_t* perm, int64_t perm_chunk, const int64_t* xstate, int64_t xmax) {
  return sddmm_entry(in_dtype, s_dtype, idx_dtype, nnz, rows, cols, s_data, A, lda, Bt, ldb, K, out, stream, perm, perm_chunk,
                     xstate, xmax);
}

static int sddmm_entry(int in_dtype, int s_dtype, int idx_dtype, int64_t nnz, const void* rows, const void* cols,
                       const void* s_data, const void* A, int64_t lda, const void* Bt, int64_t ldb, int64_t K,
                       void* out, void* stream, const int64_t* perm, int64_t perm_chunk, const int64_t* xstate,
                       int64_t xmax) {
  if (nnz < 0 || K < 0) return SPAMD_EINVAL;
  if (nnz == 0) return 0;
  if (((uintptr_t)A % 16) || ((uintptr_t)Bt % 16)) return SPAMD_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int esz = in_dtype == SPAMD_BF16 ? 2 : (in_dtype == SPAMD_F32 ? 4 : 8);
  if ((lda * esz) % 16 || (ldb * esz) % 16) return SPAMD_EINVAL;
  SPAMD_DISPATCH_IDX(idx_dtype, I, {
    const I* r = (const I*)rows;
    const I* c = (const I*)cols;
    if (in_dtype == SPAMD_BF16 && s_dtype == SPAMD_F32)
      return launch_sddmm<__hip_bfloat16, float, I>(nnz, r, c, (const float*)s_data, (const __hip_bfloat16*)A, lda,
                                                    (const __hip_bfloat16*)Bt, ldb, K, (float*)out, st, perm, perm_chunk, xstate, xmax);
    if (in_dtype == SPAMD_F32 && s_dtype == SPAMD_F32)
      return launch_sddmm<float, float, I>(nnz, r, c, (const float*)s_data, (const float*)A, lda, (const float*)Bt,
                                           ldb, K, (float*)out, st, perm, perm_chunk, xstate, xmax);
    if (in_dtype == SPAMD_F64 && s_dtype == SPAMD_F64)
      return launch_sddmm<double, double, I>(nnz, r, c, (const double*)s_data, (const double*)A, lda,
                                             (const double*)Bt, ldb, K, (double*)out, st, perm, perm_chunk, xstate, xmax);
  })
  return SPAMD_ETYPE;
}
