// Pieces shared by the SDDMM kernels (sddmm.hip: gather and row-major row-cached kernels; sddmm_panel.hip: the
// column-panel kernel): accumulator type, the per-lane dot product, the lane-group sum and broadcast.
#pragma once
#include "common.h"
#include <hip/hip_bf16.h>

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

typedef __bf16 sd_bf2 __attribute__((ext_vector_type(2)));

// <a, b> over KS 16-byte vectors per lane.  bf16: v_dot2c_f32_bf16 (two products and the add per instruction, fp32
// accumulate; no bf16 -> fp32 conversions); fp32/fp64: fused multiply-adds.
template <typename TIN, typename VT, int KS>
__device__ __forceinline__ typename Acc<TIN>::type sd_dot(const VT (&av)[KS], const VT (&bv)[KS]) {
  using ACC = typename Acc<TIN>::type;
  constexpr int EPL = 16 / (int)sizeof(TIN);
  ACC acc = 0;
  if constexpr (std::is_same<TIN, __hip_bfloat16>::value) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      sd_bf2 a2[4], b2[4];
      __builtin_memcpy(a2, &av[s], 16);
      __builtin_memcpy(b2, &bv[s], 16);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_fdot2_f32_bf16(a2[e], b2[e], acc, false);
    }
  } else {
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int e = 0; e < EPL; ++e) acc = __builtin_fma(to_acc(av[s].v[e]), to_acc(bv[s].v[e]), acc);
  }
  return acc;
}

template <int LPN, typename ACC>
__device__ __forceinline__ ACC sd_group_sum(ACC acc);

// The dot product of a lane group in the order every kernel of the family uses.  Rows of exactly 1 KB (LPN = 16, KS = 4) are
// summed as TWO 512-byte halves - each half's 16-lane sum, then first + second - because the column-panel order computes them
// in two passes over half-rows (spamd_sddmm_panels); every other row length is one sum over the lanes' whole shares.
template <typename TIN, typename VT, int LPN, int KS>
__device__ __forceinline__ typename Acc<TIN>::type sd_dot_group(const VT (&av)[KS], const VT (&bv)[KS]);

template <int CTRL>
__device__ __forceinline__ float sd_dpp(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}

// Sum over the LPN (>= 16) lanes of a group, left in every lane.  fp32: the first 16 lanes in four DPP adds
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror), wider groups finish with shuffles.
template <int LPN, typename ACC>
__device__ __forceinline__ ACC sd_group_sum(ACC acc) {
  static_assert(LPN >= 16, "a DPP row is 16 lanes");
  if constexpr (sizeof(ACC) == 4) {
    acc += sd_dpp<0xB1>(acc);
    acc += sd_dpp<0x4E>(acc);
    acc += sd_dpp<0x141>(acc);
    acc += sd_dpp<0x140>(acc);
#pragma unroll
    for (int off = 16; off < LPN; off <<= 1) acc += __shfl_xor(acc, off, 64);
  } else {
#pragma unroll
    for (int off = LPN / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  }
  return acc;
}

template <typename TIN, typename VT, int LPN, int KS>
__device__ __forceinline__ typename Acc<TIN>::type sd_dot_group(const VT (&av)[KS], const VT (&bv)[KS]) {
  if constexpr (LPN == 16 && KS == 4) {
    const VT(&a0)[2] = reinterpret_cast<const VT(&)[2]>(av[0]);
    const VT(&a1)[2] = reinterpret_cast<const VT(&)[2]>(av[2]);
    const VT(&b0)[2] = reinterpret_cast<const VT(&)[2]>(bv[0]);
    const VT(&b1)[2] = reinterpret_cast<const VT(&)[2]>(bv[2]);
    const auto first = sd_group_sum<LPN>(sd_dot<TIN, VT, 2>(a0, b0));
    const auto second = sd_group_sum<LPN>(sd_dot<TIN, VT, 2>(a1, b1));
    return first + second;
  } else {
    return sd_group_sum<LPN>(sd_dot<TIN, VT, KS>(av, bv));
  }
}

// Lane `U` of every 16-lane row -> all lanes of that row (v_mov_b32_dpp row_newbcast:U); wider groups go through a
// shuffle.  4- and 8-byte values.
template <int LPN, int U, typename T>
__device__ __forceinline__ T sd_bcast(T x) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "4- or 8-byte values");
  if constexpr (LPN == 16) {
    if constexpr (sizeof(T) == 4) {
      return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x150 + U, 0xf, 0xf, false));
    } else {
      const uint64_t b = __builtin_bit_cast(uint64_t, x);
      const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, 0x150 + U, 0xf, 0xf, false);
      const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), 0x150 + U, 0xf, 0xf, false);
      return __builtin_bit_cast(T, ((uint64_t)hi << 32) | lo);
    }
  } else {
    return __shfl(x, U, LPN);
  }
}

// Elements U0 .. U0+3 of a step (see sddmm_rowcache_kernel): the four Bt rows are requested first, then each element
// is finished in turn; the A row is (re)loaded only when the element's row differs from the one in registers.
// 16 bytes with the non-temporal hint (`global_load_dwordx4 ... nt`)
template <typename VT>
__device__ __forceinline__ VT sd_load_nt(const void* p) {
  static_assert(sizeof(VT) == 16, "16-byte vectors");
  typedef unsigned sd_u4 __attribute__((ext_vector_type(4)));
  const sd_u4 x = __builtin_nontemporal_load(reinterpret_cast<const sd_u4*>(p));
  VT r;
  __builtin_memcpy(&r, &x, 16);
  return r;
}

// 16 bytes from memory, waited for INSIDE the statement (`s_waitcnt vmcnt(0)`), as one opaque block: for a load on a rare arm
// of a branch whose other arm fills the same registers from LDS.  As a plain load the compiler has to assume after the join
// that the registers are still in flight and puts a full memory wait in front of their first use on BOTH arms - every load
// issued before it (the loads the common arm wants to keep in flight) is waited for too (round 6, tools/isa_wait_audit.py).
template <typename VT>
__device__ __forceinline__ VT sd_load_now(const void* p) {
  static_assert(sizeof(VT) == 16, "16-byte vectors");
  typedef unsigned sd_u4 __attribute__((ext_vector_type(4)));
  sd_u4 x;
  asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(p) : "memory");
  VT r;
  __builtin_memcpy(&r, &x, 16);
  return r;
}

}  // namespace spamd
