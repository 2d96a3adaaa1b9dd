// A9: sampled dense-dense matrix multiplication,  out[n] = s[n] * sum_k A[i_n, k] * Bt[j_n, k].
//
// The reference has no SDDMM kernel: `s * (a @ b)` (examples/sddmm_example.py:51-52) forms the
// full dense M x N product with BLAS and then gathers it at the mask's coordinates
// (`_Elemwise`, _umath.py:602-633) — 10^10 elements for BASELINE config 4.  Here only the
// sampled dot products are formed: LPN lanes of a wave own one stored element, stream the two
// K-long rows with 16-byte loads (B is taken K-major, i.e. as Bt = B^T row-major, so both rows
// are contiguous), multiply-accumulate in fp32 (bf16/fp32 inputs) or fp64, and reduce across
// the LPN lanes with wave shuffles.  Gather-bound (L2): 2*K*sizeof(in) bytes per element.
#include "common.h"
#include <hip/hip_bf16.h>
#include <stdlib.h>

namespace spamd {

template <typename TIN>
struct Acc { using type = float; };
template <>
struct Acc<double> { using type = double; };

template <typename TIN>
__device__ __forceinline__ typename Acc<TIN>::type to_acc(TIN x) {
  if constexpr (std::is_same<TIN, __hip_bfloat16>::value) return __bfloat162float(x);
  else return (typename Acc<TIN>::type)x;
}

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

// Row-cached form for masks stored in row-major order (canonical COO / CSR order): a lane group walks a
// CONTIGUOUS chunk of stored elements, so the A row of consecutive elements is usually the same one and stays
// in registers (KS vectors per lane); only the Bt rows are gathered — half the L1 traffic of the kernel above,
// which is what bounds it (2 * K * sizeof(in) bytes per element through a 64 B/clk/CU path).
// K == LPN * KS * EPL exactly.  UNR elements are in flight; a batch that straddles a row change takes the
// per-element path.  Any element order is correct; only row-major order is fast.
template <typename TIN, typename TS, typename I, int LPN, int KS, int UNR>
__global__ void __launch_bounds__(256)
sddmm_rowcache_kernel(int64_t nnz, int64_t chunk, const I* __restrict__ rows, const I* __restrict__ cols,
                      const TS* __restrict__ s_data, const TIN* __restrict__ A, int64_t lda,
                      const TIN* __restrict__ Bt, int64_t ldb, TS* __restrict__ out) {
  using ACC = typename Acc<TIN>::type;
  constexpr int EPL = 16 / (int)sizeof(TIN);
  using VT = Vec<TIN, EPL>;
  const int sub = (threadIdx.x & 63) % LPN;
  const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPN;
  const int64_t nbeg = group * chunk;
  const int64_t nend = nbeg + chunk < nnz ? nbeg + chunk : nnz;
  const int koff = sub * EPL;
  int64_t cur = -1;
  VT av[KS];
  auto load_a = [&](int64_t r) {
    const TIN* ar = A + r * lda + koff;
#pragma unroll
    for (int s = 0; s < KS; ++s) av[s] = *reinterpret_cast<const VT*>(ar + s * LPN * EPL);
  };
  auto dot = [&](const VT (&bv)[KS]) {
    ACC acc = 0;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int e = 0; e < EPL; ++e) acc = __builtin_fma(to_acc(av[s].v[e]), to_acc(bv[s].v[e]), acc);
    return acc;
  };
  auto finish = [&](ACC acc, int64_t n) {
#pragma unroll
    for (int off = LPN / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (sub == 0) out[n] = (TS)((ACC)s_data[n] * acc);
  };
  // (all lane groups of a wave run the same number of iterations: __shfl_xor needs every lane of a group,
  // and groups only differ in whether they have work left, which is uniform inside a group)
  for (int64_t n0 = nbeg; n0 < nend; n0 += UNR) {
    int64_t r[UNR];
    const TIN* br[UNR];
    bool same = true;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int64_t n = n0 + u < nend ? n0 + u : nend - 1;
      r[u] = (int64_t)rows[n];
      br[u] = Bt + (int64_t)cols[n] * ldb + koff;
      same = same && r[u] == r[0];
    }
    VT bv[UNR][KS];
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
      for (int s = 0; s < KS; ++s) bv[u][s] = *reinterpret_cast<const VT*>(br[u] + s * LPN * EPL);
    if (same) {
      if (r[0] != cur) {
        cur = r[0];
        load_a(cur);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u)
        if (n0 + u < nend) finish(dot(bv[u]), n0 + u);
    } else {
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (r[u] != cur) {
          cur = r[u];
          load_a(cur);
        }
        if (n0 + u < nend) finish(dot(bv[u]), n0 + u);
      }
    }
  }
}

template <typename TIN, typename TS, typename I>
static int launch_sddmm(int64_t nnz, const I* rows, const I* cols, const TS* s, const TIN* A, int64_t lda,
                        const TIN* Bt, int64_t ldb, int64_t K, TS* out, hipStream_t st) {
  constexpr int EPL = 16 / (int)sizeof(TIN);
  const int64_t vecs = K / EPL;
  int lpn = 4;
  while (lpn < 64 && vecs > lpn * 2) lpn <<= 1;  // ~2 vector loads per lane per operand
  // measured on MI355X (config 4): bf16 rows (512 B) are fastest with one element per lane group
  // in flight (0.86 ms), fp32/fp64 rows with four (1.44 ms vs 1.54 ms)
  {
    // row-cached kernel: K must be LPN * KS vectors exactly (KS <= 4); try 16 lanes per element first
    const char* v = getenv("SPAMD_SDDMM_VARIANT");  // tuning hook: "0" = gather kernel only
    const bool allow = !(v && v[0] == '0');
    for (int L = 16; allow && L <= 64; L <<= 1) {
      if (vecs % L) continue;
      const int ks = (int)(vecs / L);
      if (ks < 1 || ks > 4 || ks == 3) continue;
      constexpr int U = 4;
      const int64_t groups_wanted = 256 * 16 * (256 / L);  // 16 workgroups per CU
      int64_t chunk = ceil_div(nnz, groups_wanted);
      chunk = ceil_div(chunk, (int64_t)U) * U;
      const int64_t groups = ceil_div(nnz, chunk);
      const int64_t blocks = ceil_div(groups * L, (int64_t)256);
#define SDR(LL, KK)                                                                                           \
  if (L == LL && ks == KK) {                                                                                  \
    hipLaunchKernelGGL((sddmm_rowcache_kernel<TIN, TS, I, LL, KK, U>), dim3((unsigned)blocks), dim3(256), 0, st, \
                       nnz, chunk, rows, cols, s, A, lda, Bt, ldb, out);                                      \
    return launch_status();                                                                                   \
  }
      SDR(16, 1) SDR(16, 2) SDR(16, 4) SDR(32, 1) SDR(32, 2) SDR(32, 4) SDR(64, 1) SDR(64, 2) SDR(64, 4)
#undef SDR
    }
  }
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

extern "C" int spamd_sddmm(int in_dtype, int s_dtype, int idx_dtype, int64_t nnz, const void* rows, const void* cols,
                           const void* s_data, const void* A, int64_t lda, const void* Bt, int64_t ldb, int64_t K,
                           void* out, void* stream) {
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
                                                    (const __hip_bfloat16*)Bt, ldb, K, (float*)out, st);
    if (in_dtype == SPAMD_F32 && s_dtype == SPAMD_F32)
      return launch_sddmm<float, float, I>(nnz, r, c, (const float*)s_data, (const float*)A, lda, (const float*)Bt,
                                           ldb, K, (float*)out, st);
    if (in_dtype == SPAMD_F64 && s_dtype == SPAMD_F64)
      return launch_sddmm<double, double, I>(nnz, r, c, (const double*)s_data, (const double*)A, lda,
                                             (const double*)Bt, ldb, K, (double*)out, st);
  })
  return SPAMD_ETYPE;
}
