// Shared device/host helpers for libsparse_amd (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

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

inline int launch_status() {
  hipError_t e = hipGetLastError();
  return (int)e;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

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
