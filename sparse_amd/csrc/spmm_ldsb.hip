// A1/A3 with a SHORT contracted axis: CSR x dense -> dense with the dense operand resident in LDS.
// (`_dot_csr_ndarray` / `_dot_coo_ndarray`, reference sparse/numba_backend/_common.py:720-755, 979-1014; BASELINE config 3:
// tensordot of a (512, 512, 512) COO with a 512 x 512 matrix = 262144 rows of ~5 stored elements, K = 512, N = 512.)
//
// With few stored elements per row the product is all output writes plus, per stored element, a whole row of B: through the
// vector-memory path those rows are 2.7 GB (fp32) of gathers for 0.5 GB of output.  When K is small a column
// PANEL of B fits in LDS: 16 lanes x 16 bytes = 256 bytes per row, K rows <= 144 KB (K <= 576).  A workgroup copies its
// panel of B once and then streams rows of A: 16 lanes own a row (4 rows per wave), the row's first 16 (index, value) pairs are
// fetched with one coalesced load and handed round with `row_newbcast` DPP moves (no LDS traffic, no SALU), four elements
// at a time without a per-element predicate (see `neutral_pad`), every element
// costs one `ds_read_b128` of the B row and VEC multiply-adds, and the 256-byte result segment leaves with one
// non-temporal store.  The sum of an output element runs over the row's elements in storage order by ONE lane: the
// reference's order, bit-identical to it under SPAMD_EXACT_MULADD, identical to the row-group kernel's either way.
// The row chain (pointers -> elements -> LDS -> store) is software-pipelined over the wave's row slots as in the row-vector
// kernel of spmm_csr.hip.  A is re-read once per panel (N / 64 times for fp32): it is the small operand here.
#include "common.h"
#include <mutex>

namespace spamd {

constexpr int LB_LANES = 16;                 // lanes per row: one 256-byte panel row
constexpr int LB_ROW_BYTES = 256;
constexpr int LB_LDS_BYTES = 160 * 1024;     // all of the CU's LDS (144 KB until late round 4)

template <int J, typename T>
__device__ __forceinline__ T row_bcast(T x) {
  // lane J of every 16-lane row to all lanes of that row (DPP row_newbcast, gfx90a+; every lane is written, so the
  // destination's previous value does not matter: `mov_dpp` leaves it undefined and costs no initialising move)
  if constexpr (sizeof(T) == 4) {
    const int r = __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x150 + J, 0xf, 0xf, false);
    return __builtin_bit_cast(T, r);
  } else {
    const long long v = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_mov_dpp((int)(v & 0xffffffffll), 0x150 + J, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp((int)(v >> 32), 0x150 + J, 0xf, 0xf, false);
    const long long r = ((long long)hi << 32) | (unsigned int)lo;
    return __builtin_bit_cast(T, r);
  }
}

// The additive identity that survives a multiplication by +0: a row of B made of these stands in for "no element" (the
// lanes of a 16-pair chunk beyond the row's end carry value +0 and this row's address), so the hand-round below needs no
// per-element predicate: (+0) * (-0.0) = -0.0 and x + (-0.0) = x for every x, -0.0 and +0.0 included; integers: 0.
template <typename T>
__device__ __forceinline__ T neutral_pad() {
  if constexpr (std::is_floating_point<T>::value) return -T(0);
  else return T(0);
}

template <typename T, typename I, bool EXACT>
__global__ void __launch_bounds__(1024)
spmm_csr_ldsb_kernel(int64_t M, int64_t K, int64_t N, const T* __restrict__ a_data, const I* __restrict__ a_idx,
                     const I* __restrict__ a_ptr, const T* __restrict__ b, int64_t ldb, T* __restrict__ out, int64_t ldo) {
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int PW = LB_LANES * VEC;          // panel width in elements
  constexpr int RPW = SPAMD_WAVE / LB_LANES;  // rows per wave
  using V = Vec<T, VEC>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lb_smem[];
  V* bl = reinterpret_cast<V*>(lb_smem);      // [K + 1][16] vectors; row K is the neutral row

  const int64_t c0 = (int64_t)blockIdx.y * PW;
  const int lane = threadIdx.x & (SPAMD_WAVE - 1);
  const int gl = lane & (LB_LANES - 1);
  const int sub = lane / LB_LANES;
  const int64_t col = c0 + (int64_t)gl * VEC;
  const bool col_ok = col < N;                // N % VEC == 0 (dispatcher)
  const int lane_off = gl * 16;               // this lane's 16 bytes of a panel row
  const int pad_row = (int)K * LB_ROW_BYTES;

  // the panel of B: K rows of 256 bytes, consecutive threads on consecutive 16-byte pieces
  for (int64_t i = threadIdx.x; i < (K + 1) * LB_LANES; i += blockDim.x) {
    const int64_t k = i / LB_LANES;
    const int64_t c = c0 + (i % LB_LANES) * VEC;
    V v;
#pragma unroll
    for (int e = 0; e < VEC; ++e) v.v[e] = k < K ? T(0) : neutral_pad<T>();
    if (k < K && c < N) v = *reinterpret_cast<const V*>(b + k * ldb + c);
    bl[i] = v;
  }
  __syncthreads();

  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x / SPAMD_WAVE) + (threadIdx.x / SPAMD_WAVE);
  const int64_t stride = (int64_t)gridDim.x * (blockDim.x / SPAMD_WAVE) * RPW;
  // Loads are issued one (elements) and two (row pointers) slots ahead and only CONSUMED in the slot they belong to:
  // every load below is unconditional (clamped address, validity applied at use) and nothing is computed from a loaded
  // value in the iteration that issues it - a branch around a load or a shift of its result makes the compiler wait for
  // the load on the spot, which serialises the chain again (measured: 0.159 -> see DESIGN.md A3).
  const int64_t nnz = uniform((int64_t)a_ptr[M]);
  if (nnz == 0) {      // nothing stored (a_indices / a_data may be null): zeros
    T zero[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) zero[e] = T(0);
    for (int64_t r = wave * RPW + sub; r < M; r += stride)
      if (col_ok) nt_store<T, VEC>(out + r * ldo + col, zero);
    return;
  }
  struct Raw {
    I ci;
    T vi;
  };
  // Row numbers and element positions are computed in the index type's own width: with 32-bit indices nnz and M are below
  // 2^31, so unsigned 32-bit arithmetic (one instruction per step instead of the carry chains and paired selects of 64-bit
  // values) is exact; only the final address is 64 bits wide.
  using U = typename std::conditional<sizeof(I) == 4, unsigned, int64_t>::type;
  const U m_rows = (U)M, n_el = (U)nnz;
  const U ustride = (U)stride;
  auto ptrs = [&](U rbase, I& s, I& e) {
    U r = rbase + (U)sub;
    r = r < m_rows ? r : m_rows - 1;
    s = a_ptr[r];
    e = a_ptr[r + 1];
  };
  auto head = [&](U s, Raw& h) {
    U p = s + (U)gl;
    p = p < n_el ? p : n_el - 1;
    h.ci = a_idx[p];
    h.vi = a_data[p];
  };
  U base = (U)(wave * RPW);
  I ps0, pe0, ps1, pe1, ps2, pe2;
  Raw cur, nxt;
  ptrs(base, ps0, pe0);
  ptrs(base + ustride, ps1, pe1);
  head((U)ps0, cur);
  for (; base < m_rows; base += ustride) {
    ptrs(base + 2 * ustride, ps2, pe2);
    head((U)ps1, nxt);
    T acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = T(0);
    const bool row_ok = base + (U)sub < m_rows;
    const U s0 = (U)ps0;
    const U len = row_ok ? (U)pe0 - s0 : (U)0;
#define SPAMD_LB_STEP4(J)                                                                                     \
  {                                                                                                           \
    const int a0 = row_bcast<J>(ca) + lane_off, a1 = row_bcast<J + 1>(ca) + lane_off;                         \
    const int a2 = row_bcast<J + 2>(ca) + lane_off, a3 = row_bcast<J + 3>(ca) + lane_off;                     \
    const V b0 = *reinterpret_cast<const V*>(lb_smem + a0), b1 = *reinterpret_cast<const V*>(lb_smem + a1);   \
    const V b2 = *reinterpret_cast<const V*>(lb_smem + a2), b3 = *reinterpret_cast<const V*>(lb_smem + a3);   \
    const T v0 = row_bcast<J>(va), v1 = row_bcast<J + 1>(va), v2 = row_bcast<J + 2>(va),                      \
            v3 = row_bcast<J + 3>(va);                                                                        \
    _Pragma("unroll") for (int e = 0; e < VEC; ++e) acc[e] = mul_add<EXACT>(v0, b0.v[e], acc[e]);             \
    _Pragma("unroll") for (int e = 0; e < VEC; ++e) acc[e] = mul_add<EXACT>(v1, b1.v[e], acc[e]);             \
    _Pragma("unroll") for (int e = 0; e < VEC; ++e) acc[e] = mul_add<EXACT>(v2, b2.v[e], acc[e]);             \
    _Pragma("unroll") for (int e = 0; e < VEC; ++e) acc[e] = mul_add<EXACT>(v3, b3.v[e], acc[e]);             \
  }
    // one chunk of up to 16 pairs per row: `h` holds this lane's pair, `done` pairs of the row came before
    auto chunk = [&](const Raw& h, U done) {
      const U left = len > done ? len - done : (U)0;      // (U may be unsigned: no negative remainders)
      const int cnt = (int)(left < (U)LB_LANES ? left : (U)LB_LANES);
      const bool mine = gl < cnt;
      const int ca = mine ? (int)h.ci * LB_ROW_BYTES : pad_row;   // LDS byte address of the B row of this lane's element
      const T va = mine ? h.vi : T(0);
      // the longest of the wave's four rows decides how many groups of four elements are handed round
      const int wmax = max(max(__builtin_amdgcn_readlane(cnt, 0), __builtin_amdgcn_readlane(cnt, 16)),
                           max(__builtin_amdgcn_readlane(cnt, 32), __builtin_amdgcn_readlane(cnt, 48)));
#if !defined(LB_ABL) || LB_ABL != 1      // (ablation builds, tools/build_variant.sh: 1 = no hand-round, 2 = no store)
      if (wmax > 0) SPAMD_LB_STEP4(0)
      if (wmax > 4) SPAMD_LB_STEP4(4)
      if (wmax > 8) SPAMD_LB_STEP4(8)
      if (wmax > 12) SPAMD_LB_STEP4(12)
#else
      acc[0] += (T)wmax + va + (T)ca;
#endif
    };
    // the first chunk stands outside the loop over longer rows: a loop header here would merge "pair from the prefetch"
    // with "pair just loaded" and make the compiler wait for every outstanding load, the prefetches of this trip included
    chunk(cur, (U)0);
    if (__any(len > (U)LB_LANES ? 1 : 0)) {
      // rows longer than 16 (every lane of the wave takes part in the DPP moves, so the loop runs while ANY of the four
      // rows has pairs left)
      for (U done = LB_LANES; __any(done < len ? 1 : 0); done += LB_LANES) {
        Raw h;
        head(s0 + done, h);
        chunk(h, done);
      }
    }
#undef SPAMD_LB_STEP4
#if defined(LB_ABL) && LB_ABL == 2
    if (row_ok && col_ok && acc[0] == T(12345.678)) nt_store<T, VEC>(out + (int64_t)(base + (U)sub) * ldo + col, acc);
#else
    if (row_ok && col_ok) nt_store<T, VEC>(out + (int64_t)(base + (U)sub) * ldo + col, acc);
#endif
    ps0 = ps1; pe0 = pe1; ps1 = ps2; pe1 = pe2;
    cur = nxt;
  }
}

template <typename T, typename I, bool EXACT>
static int launch_ldsb(int64_t M, int64_t K, int64_t N, const T* a_data, const I* a_idx, const I* a_ptr, const T* b, int64_t ldb,
                       T* out, int64_t ldo, hipStream_t s) {
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int PW = LB_LANES * VEC;
  const size_t ldsbytes = (size_t)(K + 1) * LB_ROW_BYTES;      // (+ the neutral row)
  const unsigned panels = (unsigned)ceil_div(N, (int64_t)PW);
  // one 1024-thread workgroup per CU while the panel takes more than half of the LDS, two otherwise
  const int per_cu = ldsbytes > 80 * 1024 ? 1 : 2;
  int64_t per_panel = ceil_div((int64_t)256 * per_cu, (int64_t)panels);
  const int64_t slots = ceil_div(M, (int64_t)4 * 16);     // row slots of a 16-wave workgroup
  if (per_panel > slots) per_panel = slots;
  if (per_panel < 1) per_panel = 1;
  auto kern = spmm_csr_ldsb_kernel<T, I, EXACT>;
  if (set_max_dynamic_lds((const void*)kern, LB_LDS_BYTES)) return SPAMD_EINVAL;
  hipLaunchKernelGGL(kern, dim3((unsigned)per_panel, panels), dim3(1024), ldsbytes, s, M, K, N, a_data, a_idx, a_ptr, b, ldb,
                     out, ldo);
  return launch_status();
}

}  // namespace spamd

// 1 when spamd_spmm_csr_ldsb covers the shapes and alignments (the policy of spamd_spmm_csr's dispatcher)
extern "C" int spamd_spmm_csr_ldsb_fits(int val_dtype, int64_t M, int64_t K, int64_t N, const void* b, int64_t ldb,
                                        const void* out, int64_t ldo) {
  const int64_t es = (val_dtype == SPAMD_F64 || val_dtype == SPAMD_I64) ? 8 : 4;
  const int64_t vec = 16 / es;
  if (K < 1 || (K + 1) * spamd::LB_ROW_BYTES > spamd::LB_LDS_BYTES) return 0;
  if (N % vec || ldb % vec || ldo % vec || ((uintptr_t)b % 16) || ((uintptr_t)out % 16)) return 0;
  return 1;
}

extern "C" int spamd_spmm_csr_ldsb(int val_dtype, int idx_dtype, int64_t M, int64_t K, int64_t N, const void* a_data,
                                   const void* a_indices, const void* a_indptr, const void* b, int64_t ldb, void* out,
                                   int64_t ldo, unsigned flags, void* stream) {
  using namespace spamd;
  if (M < 0 || K < 0 || N < 0) return SPAMD_EINVAL;
  if (M == 0 || N == 0) return 0;
  if (!a_indptr || !out || !b || ldo < N || ldb < N) return SPAMD_EINVAL;
  if (!spamd_spmm_csr_ldsb_fits(val_dtype, M, K, N, b, ldb, out, ldo)) return SPAMD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const bool exact = (flags & SPAMD_EXACT_MULADD) != 0;
  SPAMD_DISPATCH_VAL(val_dtype, T, {
    SPAMD_DISPATCH_IDX(idx_dtype, I, {
      const T* ad = (const T*)a_data;
      const I* ai = (const I*)a_indices;
      const I* ap = (const I*)a_indptr;
      if constexpr (std::is_floating_point<T>::value) {
        if (exact) return launch_ldsb<T, I, true>(M, K, N, ad, ai, ap, (const T*)b, ldb, (T*)out, ldo, s);
      }
      return launch_ldsb<T, I, false>(M, K, N, ad, ai, ap, (const T*)b, ldb, (T*)out, ldo, s);
    })
  })
  return SPAMD_ETYPE;
}
