// A1, LDS-DMA streaming form: CSR x dense -> dense (reference loop:
// sparse/numba_backend/_common.py:744-753).
//
// Measured on MI355X (tools/micro/gather_bw.hip): gathering random 512-byte rows of a 5 MB
// table runs at 18.5 TB/s with 8-byte-per-lane loads, but 25 TB/s with 16-byte-per-lane loads
// or LDS-DMA (31 TB/s when the table fits the 4 MiB L2) — the vector-memory front end is the
// limiter, so the kernel must (a) move 16 B per lane per instruction and (b) never let the
// gather queue drain.  Design:
//   * a wave owns RB consecutive rows and walks their stored elements as ONE flat stream
//     (no pipeline drain at row ends; a row end costs one store of the accumulators);
//   * every B row travels HBM/L2 -> LDS by `global_load_lds_dwordx4` (16 B per lane, so one
//     instruction fetches EPI = 1024/(N*sizeof(T)) rows: 2 rows of 128 fp32) into a per-wave
//     ring of DEPTH 1-KiB slots, DEPTH*EPI elements ahead of the multiply-adds.  No VGPR is
//     ever "in flight", so the compiler cannot copy a half-loaded register, and the only
//     synchronisation is a hand-placed `s_waitcnt vmcnt(DEPTH-1)` before a slot is read:
//     LDS-DMA ops are loads, loads return in order, so once at most DEPTH-1 ops are
//     outstanding the oldest slot has landed (younger index/value DMAs or row stores can only
//     make the wait longer);
//   * the (index, value) stream itself is DMA'd 64 elements at a time into a 128-element
//     circular LDS buffer; the issue side reads its column index from there with one
//     ds_read (two distinct addresses per wave -> broadcast), the consume side its value;
//   * lane l owns VEC contiguous output columns and accumulates them strictly in storage
//     order: deterministic, and bit-identical to the reference under SPAMD_EXACT_MULADD;
//   * `out` is written once, non-temporally, from inline asm (see hidden_nt_store).
#include "spmm_internal.h"

namespace spamd {

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)p;  // LDS (address space 3) pointers are 32-bit offsets
}

// One LDS-DMA instruction: every active lane copies 16 B from `src` to LDS at
// m0_base + lane*16.  Invisible to the compiler's s_waitcnt pass (by design).
__device__ __forceinline__ void dma16(unsigned lds_base, const void* src) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
               :
               : "s"(lds_base), "v"(src)
               : "memory", "m0");
}
// 4 B per lane: lane i -> LDS m0_base + i*4.
__device__ __forceinline__ void dma4(unsigned lds_base, const void* src) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off"
               :
               : "s"(lds_base), "v"(src)
               : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}

template <typename T, typename I, int EPI, int DEPTH, bool EXACT>
__global__ void __launch_bounds__(256)
spmm_csr_ldsring_kernel(int64_t M, int64_t N, const T* __restrict__ a_data,
                        const I* __restrict__ a_idx, const I* __restrict__ a_ptr,
                        const T* __restrict__ b, int64_t ldb, T* __restrict__ out, int64_t ldo,
                        int RB /* rows per task, <= 64 */) {
  constexpr int W = SPAMD_WAVE;
  constexpr int LPE = W / EPI;                          // lanes that fetch one B row segment
  constexpr int SEG = 16 * LPE;                         // bytes of a B row fetched per element
  constexpr int SEGN = SEG / (int)sizeof(T);            // columns per pass
  constexpr int VEC = SEGN / W;                         // columns owned by a lane
  static_assert(VEC >= 1, "segment too narrow for this dtype");
  static_assert(DEPTH >= 2 && (32 % DEPTH) == 0, "DEPTH must divide 32");
  constexpr int CH = 64;                                // stream elements per index/value DMA
  constexpr int IW = (int)sizeof(I) / 4, TW = (int)sizeof(T) / 4;
  using V = Vec<T, VEC>;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WAVE_LDS = DEPTH * 1024 + 2 * CH * (int)sizeof(I) + 2 * CH * (int)sizeof(T);
  const int lane = threadIdx.x & (W - 1);
  const int wv = uniform((int)(threadIdx.x / W));
  char* const ring = smem + wv * WAVE_LDS;
  I* const cols = reinterpret_cast<I*>(ring + DEPTH * 1024);              // [2*CH] circular
  T* const vals = reinterpret_cast<T*>(ring + DEPTH * 1024 + 2 * CH * sizeof(I));

  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x / W) + wv;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x / W);
  const int64_t ntasks = (M + RB - 1) / RB;
  const int sub = lane / LPE;       // which element of a slot this lane fetches
  const int sl = lane % LPE;        // its 16-byte piece of the segment

  for (int64_t task = wave; task < ntasks; task += nwaves) {
    const int64_t r0 = task * RB;
    const int nrows = (int)((M - r0) < (int64_t)RB ? (M - r0) : (int64_t)RB);
    const int64_t p_begin = uniform((int64_t)a_ptr[r0]);
    I my_end_abs = (lane < nrows) ? a_ptr[r0 + 1 + lane] : (I)0;
    const int64_t p_end = uniform((int64_t)wave_bcast(my_end_abs, nrows - 1));
    const I my_end = (I)(my_end_abs - (I)p_begin);  // lane l: relative end of row r0+l
    const int64_t total = p_end - p_begin;
    const int64_t nslots = (total + EPI - 1) / EPI;
    const T* const ad = a_data + p_begin;
    const I* const ai = a_idx + p_begin;

    for (int64_t c0 = 0; c0 < N; c0 += SEGN) {
      const char* const bseg = reinterpret_cast<const char*>(b + c0) + sl * 16;
      const int64_t col = c0 + (int64_t)lane * VEC;
      T acc[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] = T(0);
      int row = 0;
      int64_t cur_end = (int64_t)wave_bcast(my_end, 0);

      auto flush_rows = [&](int64_t pos) {
        while (row < nrows && pos == cur_end) {
          hidden_nt_store<T, VEC>(out + (r0 + row) * ldo + col, acc);
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[e] = T(0);
          ++row;
          cur_end = (row < nrows) ? (int64_t)wave_bcast(my_end, row) : (int64_t)-1;
        }
      };

      // index/value chunk `c` (stream positions 64c .. 64c+63) -> half (c & 1) of the buffers
      auto dma_chunk = [&](int64_t c) {
        const int64_t base = c * CH;
        if (base >= total) return;
        const int half = (int)(c & 1);
        const int64_t ndw_i = ((total - base) < CH ? (total - base) : (int64_t)CH) * IW;
        const int64_t ndw_t = ((total - base) < CH ? (total - base) : (int64_t)CH) * TW;
#pragma unroll
        for (int k = 0; k < IW; ++k) {
          if (lane + k * W < ndw_i)
            dma4(lds_addr(cols + half * CH) + k * W * 4,
                 reinterpret_cast<const unsigned*>(ai + base) + lane + k * W);
        }
#pragma unroll
        for (int k = 0; k < TW; ++k) {
          if (lane + k * W < ndw_t)
            dma4(lds_addr(vals + half * CH) + k * W * 4,
                 reinterpret_cast<const unsigned*>(ad + base) + lane + k * W);
        }
      };

      // slot `s` (stream positions EPI*s .. EPI*s+EPI-1) -> ring slot (s % DEPTH)
      auto issue_slot = [&](int64_t s) {
        int64_t p = s * EPI + sub;
        if (p >= total) p = total - 1;  // odd tail: fetch a valid row, never consumed
        const I c = cols[p & (2 * CH - 1)];
        dma16(lds_addr(ring) + (unsigned)(s % DEPTH) * 1024u, bseg + (int64_t)c * ldb * (int64_t)sizeof(T));
      };

      auto consume_slot = [&](int64_t s) {
        const char* slot = ring + (s % DEPTH) * 1024;
#pragma unroll
        for (int k = 0; k < EPI; ++k) {
          const int64_t p = s * EPI + k;
          if (p < total) {
            flush_rows(p);
            const T v = vals[p & (2 * CH - 1)];
            const V bv = *reinterpret_cast<const V*>(slot + k * SEG + lane * (VEC * (int)sizeof(T)));
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = mul_add<EXACT>(v, bv.v[e], acc[e]);
          }
        }
      };

      if (total > 0) {
        dma_chunk(0);
        dma_chunk(1);
        wait_vmcnt<0>();  // task start-up: both chunks landed
        const int64_t pro = nslots < DEPTH ? nslots : (int64_t)DEPTH;
        for (int64_t s = 0; s < pro; ++s) issue_slot(s);

        for (int64_t sb = 0; sb < nslots; sb += DEPTH) {
          // entering a new 64-element chunk: its half-buffer predecessor is fully consumed,
          // refill that half with the chunk after next
          if ((sb * EPI) % CH == 0 && sb > 0) dma_chunk((sb * EPI) / CH + 1);
          const bool steady = (sb + 2 * DEPTH <= nslots);
          if (steady) {
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
              wait_vmcnt<DEPTH - 1>();
              consume_slot(sb + u);
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // slot fully read before re-use
              issue_slot(sb + u + DEPTH);
            }
          } else {
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
              if (sb + u < nslots) {
                wait_vmcnt<0>();
                consume_slot(sb + u);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (sb + u + DEPTH < nslots) issue_slot(sb + u + DEPTH);
              }
            }
          }
        }
      }
      flush_rows(total);
      while (row < nrows) {
        hidden_nt_store<T, VEC>(out + (r0 + row) * ldo + col, acc);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = T(0);
        ++row;
      }
      wait_vmcnt<0>();  // nothing in flight when the LDS buffers are re-used
    }
  }
}

template <typename T, typename I, int EPI, int DEPTH, bool EXACT>
static int launch_ldsring(int64_t M, int64_t N, const T* a_data, const I* a_idx, const I* a_ptr,
                          const T* b, int64_t ldb, T* out, int64_t ldo, int RB, hipStream_t s) {
  constexpr int WAVE_LDS = DEPTH * 1024 + 2 * 64 * (int)sizeof(I) + 2 * 64 * (int)sizeof(T);
  constexpr int WPB = 4;
  const size_t lds = (size_t)WAVE_LDS * WPB;
  const int64_t ntasks = ceil_div(M, RB);
  int64_t blocks = ceil_div(ntasks, WPB);
  int per_cu = (int)(160 * 1024 / lds);
  if (per_cu > 8) per_cu = 8;
  if (per_cu < 1) per_cu = 1;
  const int64_t cap = 256 * (int64_t)per_cu;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void*>(&spmm_csr_ldsring_kernel<T, I, EPI, DEPTH, EXACT>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL((spmm_csr_ldsring_kernel<T, I, EPI, DEPTH, EXACT>), dim3((unsigned)blocks),
                     dim3(64 * WPB), lds, s, M, N, a_data, a_idx, a_ptr, b, ldb, out, ldo, RB);
  return launch_status();
}

// Shapes this kernel takes: N*sizeof(T) in {256, 512, 1024} or a multiple of 1024, 16-byte
// aligned rows.  Anything else: SPAMD_ETYPE (the caller falls back to the row-group kernel).
template <typename T, typename I, bool EXACT>
int spmm_csr_ldsring_dispatch(int64_t M, int64_t N, const T* a_data, const I* a_idx,
                              const I* a_ptr, const T* b, int64_t ldb, T* out, int64_t ldo,
                              int depth, int RB, hipStream_t s) {
  const int64_t rowbytes = N * (int64_t)sizeof(T);
  if (((uintptr_t)b % 16) || ((ldb * sizeof(T)) % 16) || ((uintptr_t)out % 16) ||
      ((ldo * sizeof(T)) % 16))
    return SPAMD_ETYPE;
  if (RB < 1) RB = 1;
  if (RB > 64) RB = 64;
  int epi = 0;
  if (rowbytes % 1024 == 0) epi = 1;
  else if (rowbytes == 512) epi = 2;
  else if (rowbytes == 256 && sizeof(T) == 4) epi = 4;
  if (!epi) return SPAMD_ETYPE;
#define SPAMD_LCASE(E, D)                                                                        \
  if (epi == E && depth == D)                                                                    \
    return launch_ldsring<T, I, E, D, EXACT>(M, N, a_data, a_idx, a_ptr, b, ldb, out, ldo, RB, s);
  SPAMD_LCASE(1, 4)
  SPAMD_LCASE(1, 8)
  SPAMD_LCASE(2, 4)
  SPAMD_LCASE(2, 8)
  SPAMD_LCASE(2, 16)
  if constexpr (sizeof(T) == 4) {
    SPAMD_LCASE(4, 4)
    SPAMD_LCASE(4, 8)
  }
#undef SPAMD_LCASE
  return SPAMD_ETYPE;
}

#define SPAMD_INST(T, I, E)                                                                       \
  template int spmm_csr_ldsring_dispatch<T, I, E>(int64_t, int64_t, const T*, const I*, const I*, \
                                                  const T*, int64_t, T*, int64_t, int, int,        \
                                                  hipStream_t);
SPAMD_INST(float, int32_t, false)
SPAMD_INST(float, int32_t, true)
SPAMD_INST(float, int64_t, false)
SPAMD_INST(float, int64_t, true)
SPAMD_INST(double, int32_t, false)
SPAMD_INST(double, int32_t, true)
SPAMD_INST(double, int64_t, false)
SPAMD_INST(double, int64_t, true)
SPAMD_INST(int32_t, int32_t, false)
SPAMD_INST(int32_t, int64_t, false)
SPAMD_INST(int64_t, int32_t, false)
SPAMD_INST(int64_t, int64_t, false)
#undef SPAMD_INST

}  // namespace spamd
