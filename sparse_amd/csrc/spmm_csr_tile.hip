// A1, K-blocked LDS-tile form (fp32, N = 128): CSR x dense -> dense (reference loop:
// sparse/numba_backend/_common.py:744-753).
//
// Why: the row-group kernel gathers every 512-byte B row through the vector L1 (measured ceiling
// 25 TB/s chip-wide for a 5 MB table -> 2.05 ms at config 2).  LDS delivers 256 B/clk/CU (5x the L1
// rate), but B (K*512 B) does not fit: so B is streamed through LDS in tiles of KB rows, and
// because the partial sums of a row must survive from tile to tile they stay in REGISTERS:
//   * a workgroup of WAVES waves owns WAVES*RW consecutive rows for the whole kernel; wave w keeps
//     acc[RW] (one float2 per lane per row: lane l owns columns 2l, 2l+1) — statically indexed,
//     the row loop is fully unrolled;
//   * B tile t (KB x 128 floats, row-major) is copied HBM/L2 -> LDS by LDS-DMA
//     (`global_load_lds_dwordx4`, 16 B per lane, asm, hand-placed vmcnt) into a double buffer while
//     tile t-1 is being consumed; one `s_barrier` per tile;
//   * each row's stored elements are walked ONCE, in storage order, across the tiles: a 64-element
//     window of (column, value) lives in a VGPR pair (lane <-> element), the cursor in an SGPR;
//     per element: v_readlane x2, one LDS address add, ds_read_b64, one (pk_)fma.  The window is
//     refilled from a prefetched shadow window (no exposed latency).
// B traffic through the L1 path drops from nnz*512 B (51 GB) to (M / (WAVES*RW)) * K*512 B
// (20 GB at 256 rows per workgroup); the 51 GB of operand reads move to LDS.
// Accumulation per output element is strictly in storage order by one lane: deterministic, and
// bit-identical to the reference under SPAMD_EXACT_MULADD.
//
// STATUS (round 1, measured on MI355X, config 2): correct (bit-exact in all parity cases) but
// 4.4 ms vs 2.6 ms for the row-group kernel — NOT the default; reachable only through
// SPAMD_SPMM_VARIANT="TILE=1,...".  Each stored element is a dependent chain (v_readlane -> scalar
// compare/branch -> LDS address -> ds_read_b64 (~128 clk) -> fma) of ~350 cycles per wave, and with
// ~1.3 elements per (row, tile) there is nothing to overlap it with inside a row; 16 waves per CU
// then give ~22 clk per element per CU.  Batching rows branch-free costs ~25-30 instructions per
// element (most (row, tile) slots are empty).  The form that can work organises the tile's
// elements as a flat list with DYNAMICALLY indexed accumulators (s_set_gpr_idx) and 8+ LDS reads
// in flight — hand-written assembly, planned for a later round.
#include "spmm_internal.h"

namespace spamd {

__device__ __forceinline__ void tile_dma16(unsigned lds_base, const void* src) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
               :
               : "s"(lds_base), "v"(src)
               : "memory", "m0");
}

template <typename I, bool EXACT, int RW, int KB, int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
spmm_csr_tile_kernel(int64_t M, int64_t K, const float* __restrict__ a_data, const I* __restrict__ a_idx,
                     const I* __restrict__ a_ptr, const float* __restrict__ b, int64_t ldb,
                     float* __restrict__ out, int64_t ldo) {
  constexpr int N = 128;
  constexpr int TILE_BYTES = KB * N * 4;
  constexpr int THREADS = WAVES * 64;
  constexpr int DMA_PER_THREAD = TILE_BYTES / 16 / THREADS;
  static_assert(TILE_BYTES % (16 * THREADS) == 0, "tile must be a whole number of DMA rounds");
  constexpr int SENT = 0x7fffffff;
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = uniform(tid >> 6);
  const int64_t row0 = ((int64_t)blockIdx.x * WAVES + wv) * RW;

  // ---- per-row state (fully unrolled: everything below is a scalar or a VGPR, never an array in memory)
  float2 acc[RW];
  int colw[RW], coln[RW];      // current / shadow window of column indices (lane <-> element)
  float valw[RW], valn[RW];
  int pos[RW];                 // cursor inside the current window (wave-uniform)
  int64_t nxt[RW], rend[RW];   // absolute position of the element after the shadow window; row end
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    acc[r] = make_float2(0.f, 0.f);
    const int64_t row = row0 + r;
    int64_t s = 0, e = 0;
    if (row < M) {
      s = (int64_t)a_ptr[row];
      e = (int64_t)a_ptr[row + 1];
    }
    s = uniform(s);
    e = uniform(e);
    const int64_t p0 = s + lane, p1 = s + 64 + lane;
    colw[r] = p0 < e ? (int)a_idx[p0] : SENT;
    valw[r] = p0 < e ? a_data[p0] : 0.f;
    coln[r] = p1 < e ? (int)a_idx[p1] : SENT;
    valn[r] = p1 < e ? a_data[p1] : 0.f;
    pos[r] = 0;
    nxt[r] = s + 128;
    rend[r] = e;
  }

  const int64_t ntiles = (K + KB - 1) / KB;
  auto issue_tile = [&](int64_t t) {
    const int64_t kb0 = t * KB;
    const unsigned buf = (unsigned)(t & 1) * TILE_BYTES;
#pragma unroll
    for (int i = 0; i < DMA_PER_THREAD; ++i) {
      const int e = (i * THREADS + tid) * 4;  // float index inside the tile
      const int r = e >> 7, c = e & 127;
      if (kb0 + r < K)
        tile_dma16((unsigned)(size_t)lds + buf + (unsigned)(i * WAVES + wv) * 1024u, b + (kb0 + r) * ldb + c);
    }
  };

  issue_tile(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int64_t t = 0; t < ntiles; ++t) {
    if (t + 1 < ntiles) issue_tile(t + 1);
    const int kb0 = (int)(t * KB);
    const int kb_end = kb0 + KB;
    // byte address of column pair `lane` of B row c in the current buffer: vbase + (c - kb0)*512
    const int vbase = (int)((t & 1) * TILE_BYTES) + lane * 8;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      while (true) {
        const int c = __builtin_amdgcn_readlane(colw[r], pos[r]);
        if (c >= kb_end) break;
        const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, valw[r]), pos[r]));
        const float2 bb = *reinterpret_cast<const float2*>(lds + (vbase + (c - kb0) * 512));
        acc[r].x = mul_add<EXACT>(v, bb.x, acc[r].x);
        acc[r].y = mul_add<EXACT>(v, bb.y, acc[r].y);
        if (++pos[r] == 64) {
          // window exhausted: the shadow window (fetched >= 64 elements ago) becomes current.
          // These are ordinary loads: hipcc then guards the window registers with s_waitcnt vmcnt(0)
          // inside the element loop, which also waits for the in-flight tile DMA.  Hiding them in
          // asm was tried and is NOT safe (the compiler copies the in-flight destination at the
          // join of the refill branch: stale data in 4 of 60 parity cases).
          colw[r] = coln[r];
          valw[r] = valn[r];
          pos[r] = 0;
          const int64_t p = nxt[r] + lane;
          coln[r] = p < rend[r] ? (int)a_idx[p] : SENT;
          valn[r] = p < rend[r] ? a_data[p] : 0.f;
          nxt[r] += 64;
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tile t+1 has landed (this wave's share)
    __syncthreads();                                   // ... everyone's share; tile t is free
  }

#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int64_t row = row0 + r;
    if (row < M) {
      float o[2] = {acc[r].x, acc[r].y};
      nt_store<float, 2>(out + row * ldo + lane * 2, o);
    }
  }
}

template <typename I, bool EXACT, int RW, int KB, int WAVES>
static int launch_tile(int64_t M, int64_t K, const float* a_data, const I* a_idx, const I* a_ptr, const float* b,
                       int64_t ldb, float* out, int64_t ldo, hipStream_t s) {
  constexpr int LDS = 2 * KB * 128 * 4;
  auto kern = &spmm_csr_tile_kernel<I, EXACT, RW, KB, WAVES>;
  if (LDS > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
  }
  const int64_t blocks = ceil_div(M, (int64_t)RW * WAVES);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(WAVES * 64), LDS, s, M, K, a_data, a_idx, a_ptr, b, ldb,
                     out, ldo);
  return launch_status();
}

// fp32, N == 128, K < 2^31, 16-byte aligned B rows; SPAMD_ETYPE otherwise (caller falls back).
template <typename I, bool EXACT>
int spmm_csr_tile_dispatch(int64_t M, int64_t K, int64_t N, const float* a_data, const I* a_idx, const I* a_ptr,
                           const float* b, int64_t ldb, float* out, int64_t ldo, int rw, int kb, hipStream_t s) {
  if (N != 128 || K >= ((int64_t)1 << 31) || ((uintptr_t)b % 16) || (ldb % 4) || ((uintptr_t)out % 8) || (ldo % 2))
    return SPAMD_ETYPE;
#define SPAMD_TCASE(R, KBV, WV)                                                                          \
  if (rw == R && kb == KBV)                                                                              \
    return launch_tile<I, EXACT, R, KBV, WV>(M, K, a_data, a_idx, a_ptr, b, ldb, out, ldo, s);
  SPAMD_TCASE(16, 128, 16)
  SPAMD_TCASE(16, 64, 16)
  SPAMD_TCASE(8, 128, 16)
  SPAMD_TCASE(16, 32, 8)
#undef SPAMD_TCASE
  return SPAMD_ETYPE;
}

#define SPAMD_INST(I, E)                                                                              \
  template int spmm_csr_tile_dispatch<I, E>(int64_t, int64_t, int64_t, const float*, const I*, const I*, \
                                            const float*, int64_t, float*, int64_t, int, int, hipStream_t);
SPAMD_INST(int32_t, false)
SPAMD_INST(int32_t, true)
SPAMD_INST(int64_t, false)
SPAMD_INST(int64_t, true)
#undef SPAMD_INST

}  // namespace spamd
