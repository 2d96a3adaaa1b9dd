// Shared device/host helpers for libsparse_amd (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <set>
#include <type_traits>
#include <utility>

#include "../../include/sparse_amd.h"

#define SPAMD_WAVE 64

namespace spamd {

// ---- wave-level helpers ------------------------------------------------------------------

// Broadcast lane `src` (wave-uniform) of a 32/64-bit value to every lane via v_readlane.
template <typename T>
__device__ __forceinline__ T wave_bcast(T x, int src) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "32/64-bit only");
  if constexpr (sizeof(T) == 4) {
    int r = __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), src);
    return __builtin_bit_cast(T, r);
  } else {
    long long v = __builtin_bit_cast(long long, x);
    int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), src);
    int hi = __builtin_amdgcn_readlane((int)(v >> 32), src);
    long long r = ((long long)hi << 32) | (unsigned int)lo;
    return __builtin_bit_cast(T, r);
  }
}

// Whole-wave reductions on DPP (VALU speed: six dependent steps of a few cycles; `__shfl_xor` is a ds_bpermute per step,
// ~100 cycles each through the LDS pipeline).  quad_perm xor 1 / xor 2, row_half_mirror, row_mirror leave every lane of a
// row of 16 with the row's result; row_bcast15 / row_bcast31 carry it into the later rows; lane 63 holds the wave's.
// All 64 lanes must be active.  The result is wave-uniform (an SGPR).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_mov(unsigned old, unsigned src) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  auto mn = [](unsigned a, unsigned b) { return a < b ? a : b; };
  v = mn(v, dpp_mov<0xB1, 0xf>(v, v));    // quad_perm [1,0,3,2]
  v = mn(v, dpp_mov<0x4E, 0xf>(v, v));    // quad_perm [2,3,0,1]
  v = mn(v, dpp_mov<0x141, 0xf>(v, v));   // row_half_mirror
  v = mn(v, dpp_mov<0x140, 0xf>(v, v));   // row_mirror
  v = mn(v, dpp_mov<0x142, 0xa>(v, v));   // row_bcast15 -> rows 1, 3
  v = mn(v, dpp_mov<0x143, 0xc>(v, v));   // row_bcast31 -> rows 2, 3
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
  v += dpp_mov<0xB1, 0xf>(0u, v);
  v += dpp_mov<0x4E, 0xf>(0u, v);
  v += dpp_mov<0x141, 0xf>(0u, v);
  v += dpp_mov<0x140, 0xf>(0u, v);
  v += dpp_mov<0x142, 0xa>(0u, v);
  v += dpp_mov<0x143, 0xc>(0u, v);
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// Inclusive scan over the wave on DPP row shifts (row_shr 1 / 2 / 4 / 8 inside the rows of 16 lanes, row_bcast15 / 31 across
// them): six VALU steps instead of six ds_bpermute round trips.  All 64 lanes must be active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_shift0(unsigned src) {   // lanes without a source read 0
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)src, CTRL, ROW_MASK, 0xf, true);
}
__device__ __forceinline__ unsigned wave_incl_scan_u32(unsigned x) {
  x += dpp_shift0<0x111, 0xf>(x);
  x += dpp_shift0<0x112, 0xf>(x);
  x += dpp_shift0<0x114, 0xf>(x);
  x += dpp_shift0<0x118, 0xf>(x);
  x += dpp_shift0<0x142, 0xa>(x);
  x += dpp_shift0<0x143, 0xc>(x);
  return x;
}

// Tell the compiler a value is wave-uniform (moves it to SGPRs; enables scalar branches).
__device__ __forceinline__ int uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ int64_t uniform(int64_t x) {
  int lo = __builtin_amdgcn_readfirstlane((int)(x & 0xffffffffll));
  int hi = __builtin_amdgcn_readfirstlane((int)(x >> 32));
  return ((int64_t)hi << 32) | (unsigned int)lo;
}

// Per-lane source (ds_bpermute): any lane may read any other lane.
template <typename T>
__device__ __forceinline__ T lane_shfl(T x, int src) {
  return __shfl(x, src, SPAMD_WAVE);
}

// a*b + c either fused (one rounding) or as the reference computes it (two roundings).
// HIP defaults to -ffp-contract=fast and `__fmul_rn(x,y)` is just `x*y`, so contraction
// is switched off both by the build flags (-ffp-contract=off) and by the pragma below;
// fusion happens only where __builtin_fma* is called explicitly.
template <bool EXACT, typename T>
__device__ __forceinline__ T mul_add(T a, T b, T c) {
#pragma clang fp contract(off)
  if constexpr (std::is_same<T, float>::value) {
    if constexpr (EXACT) { float p = a * b; return p + c; }
    else return __builtin_fmaf(a, b, c);
  } else if constexpr (std::is_same<T, double>::value) {
    if constexpr (EXACT) { double p = a * b; return p + c; }
    else return __builtin_fma(a, b, c);
  } else {
    return (T)(a * b + c);  // integers: exact (wrap-around) either way
  }
}

template <typename T, int N>
struct alignas(sizeof(T) * N) Vec {
  T v[N];
};

// Clang extended vectors: what the non-temporal builtins and 8/16-byte loads want.
template <typename T, int N>
struct ExtVec {
  typedef T type __attribute__((ext_vector_type(N)));
};
template <typename T>
struct ExtVec<T, 1> {
  typedef T type;
};

// Non-temporal (streaming) vector store / load of N contiguous elements.
template <typename T, int N>
__device__ __forceinline__ void nt_store(T* p, const T (&v)[N]) {
  if constexpr (N == 1) {
    __builtin_nontemporal_store(v[0], p);
  } else {
    typename ExtVec<T, N>::type x;
#pragma unroll
    for (int e = 0; e < N; ++e) x[e] = v[e];
    __builtin_nontemporal_store(x, reinterpret_cast<typename ExtVec<T, N>::type*>(p));
  }
}

// Global store that the compiler's s_waitcnt insertion does NOT see (inline asm).
// Why: on gfx9-family targets loads and stores share vmcnt and LLVM treats a pending store
// next to pending loads as "may return out of order", turning every later load wait into
// vmcnt(0) — which would drain a software-pipelined gather at every row end.  Hiding the
// store is safe for the compiler's load waits: vmcnt(N) with N = number of LATER LOADS still
// guarantees the awaited load has returned (loads return in order; an outstanding store can
// only make the wait longer).  The asm reads its data VGPRs at issue; the trailing s_nop covers
// the >8-byte store-data write-after-read hazard the compiler cannot see.
template <typename T, int N>
__device__ __forceinline__ void hidden_nt_store(T* p, const T (&v)[N]) {
  constexpr int BYTES = sizeof(T) * N;
  static_assert(BYTES == 4 || BYTES == 8 || BYTES == 16, "4/8/16-byte stores only");
  if constexpr (BYTES == 4) {
    unsigned x = __builtin_bit_cast(unsigned, v[0]);
    asm volatile("global_store_dword %0, %1, off nt" : : "v"(p), "v"(x) : "memory");
  } else if constexpr (BYTES == 8) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    u2 x;
    if constexpr (N == 1) {
      x = __builtin_bit_cast(u2, v[0]);
    } else {
      x[0] = __builtin_bit_cast(unsigned, v[0]);
      x[1] = __builtin_bit_cast(unsigned, v[1]);
    }
    asm volatile("global_store_dwordx2 %0, %1, off nt" : : "v"(p), "v"(x) : "memory");
  } else {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 x;
    if constexpr (N == 2) {
      typedef unsigned long long ull;
      ull a = __builtin_bit_cast(ull, v[0]), b = __builtin_bit_cast(ull, v[1]);
      x[0] = (unsigned)a; x[1] = (unsigned)(a >> 32); x[2] = (unsigned)b; x[3] = (unsigned)(b >> 32);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = __builtin_bit_cast(unsigned, v[e]);
    }
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(p), "v"(x) : "memory");
  }
}

// hipFuncAttributeMaxDynamicSharedMemorySize is state of the kernel's code object ON ONE DEVICE: opted in once per
// (device, kernel), not once per process (rounds 1-4 kept a per-process flag per instantiation: right for one process per
// GPU, wrong for a process that launches on two devices).  Returns 0 or the hipError_t.
inline int set_max_dynamic_lds(const void* kern, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;
  int dev = 0;
  if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return (int)e;
  std::lock_guard<std::mutex> lock(mu);
  const auto key = std::make_pair(dev, kern);
  if (done.count(key)) return 0;
  if (hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); e != hipSuccess) return (int)e;
  done.insert(key);
  return 0;
}

inline int launch_status() {
  hipError_t e = hipGetLastError();
  return (int)e;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

#ifdef __HIPCC__
// Decoupled look-back over one 64-bit state word per tile (zero-initialised): (1 << 62 | own total) = "aggregate
// known", (2 << 62 | inclusive prefix) = "prefix known"; the value travels inside the flag word, so no fence is
// needed.  Called by the 64 lanes of ONE wave (lane = 0..63); returns the sum of the totals of tiles 0 .. blk-1 in
// every lane and publishes this tile's inclusive prefix.  Tiles must be numbered in start order (a ticket taken with
// an atomic at tile start), so that a tile only ever waits for tiles that are already running.
__device__ __forceinline__ unsigned long long lookback_exclusive(unsigned long long* st, int64_t blk, unsigned long long tot,
                                                                 int lane) {
  const unsigned long long mask = (1ull << 62) - 1;
  if (lane == 0 && blk > 0)
    __hip_atomic_store(&st[blk], (1ull << 62) | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned long long excl = 0;
  int64_t hi = blk - 1;  // newest predecessor not yet accounted for
  while (hi >= 0) {
    const int64_t j = hi - lane;
    unsigned long long v = 2ull << 62;  // lanes past tile 0 behave like "prefix known, value 0"
    if (j >= 0) v = __hip_atomic_load(&st[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long flag = v >> 62;
    const unsigned long long have_prefix = __ballot(flag == 2);
    const unsigned long long missing = __ballot(flag == 0);
    // the window is usable up to the nearest lane with a prefix, if no lane up to there is still missing
    const int first_prefix = have_prefix ? __builtin_ctzll(have_prefix) : 64;
    const unsigned long long upto = first_prefix >= 63 ? ~0ull : ((2ull << first_prefix) - 1);
    if (missing & upto) continue;  // spin: re-read the window
    unsigned long long part = lane <= first_prefix ? (v & mask) : 0;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
    excl += part;
    if (first_prefix < 64) break;
    hi -= 64;
  }
  if (lane == 0)
    __hip_atomic_store(&st[blk], (2ull << 62) | (excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return excl;
}
#endif

// Exclusive scan of a SHORT int64 array by one workgroup of 1024 threads (out[i] = in[0] + .. + in[i-1], i < n; in == out
// allowed): a device-wide scan primitive costs ~30 us whatever the length (several launches, look-back state), which at
// config-1 sizes (a few hundred tile counts) is as much as the kernel it serves.
constexpr int SMALL_SCAN_MAX = 16384;
#ifdef __HIPCC__
static __global__ void __launch_bounds__(1024) small_exclusive_scan_kernel(const int64_t* in, int64_t* out, int n) {
  constexpr int PER = SMALL_SCAN_MAX / 1024;
  __shared__ int64_t wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int64_t v[PER];
  int64_t mine = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = tid * PER + j;
    v[j] = i < n ? in[i] : 0;
    mine += v[j];
  }
  int64_t incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int64_t y = __shfl_up(incl, off, 64);
    if (lane >= off) incl += y;
  }
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  int64_t run = incl - mine;
#pragma unroll
  for (int w = 0; w < 16; ++w)
    if (w < wv) run += wsum[w];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = tid * PER + j;
    if (i < n) out[i] = run;
    run += v[j];
  }
}
#endif

}  // namespace spamd

// Dispatch helpers: call F<T,I>(...) for runtime dtype codes.
#define SPAMD_DISPATCH_IDX(idx_dtype, I, ...)              \
  switch (idx_dtype) {                                     \
    case SPAMD_I32: { using I = int32_t; __VA_ARGS__; } break; \
    case SPAMD_I64: { using I = int64_t; __VA_ARGS__; } break; \
    default: return SPAMD_ETYPE;                           \
  }

#define SPAMD_DISPATCH_VAL(val_dtype, T, ...)              \
  switch (val_dtype) {                                     \
    case SPAMD_F32: { using T = float; __VA_ARGS__; } break;   \
    case SPAMD_F64: { using T = double; __VA_ARGS__; } break;  \
    case SPAMD_I32: { using T = int32_t; __VA_ARGS__; } break; \
    case SPAMD_I64: { using T = int64_t; __VA_ARGS__; } break; \
    default: return SPAMD_ETYPE;                           \
  }
