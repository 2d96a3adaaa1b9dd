// Internal cross-file declarations for the SpMM kernels.
#pragma once
#include "common.h"

namespace spamd {

// spmm_csr_ldsring.hip — LDS-DMA ring kernel; SPAMD_ETYPE when the shape does not fit it.
template <typename T, typename I, bool EXACT>
int spmm_csr_ldsring_dispatch(int64_t M, int64_t N, const T* a_data, const I* a_idx,
                              const I* a_ptr, const T* b, int64_t ldb, T* out, int64_t ldo,
                              int depth, int RB, hipStream_t s);

// spmm_csr_tile.hip — K-blocked LDS-tile kernel (fp32, N == 128); SPAMD_ETYPE when it does not apply.
template <typename I, bool EXACT>
int spmm_csr_tile_dispatch(int64_t M, int64_t K, int64_t N, const float* a_data, const I* a_idx, const I* a_ptr,
                           const float* b, int64_t ldb, float* out, int64_t ldo, int rw, int kb, hipStream_t s);

// spmm_csr_gidx.hip — LDS-tile kernel with gpr-indexed accumulators (fp32, N == 128, FMA mode).
template <typename I>
int spmm_csr_gidx_dispatch(int64_t M, int64_t K, int64_t N, const float* a_data, const I* a_idx, const I* a_ptr,
                           const float* b, int64_t ldb, float* out, int64_t ldo, int mode, hipStream_t s);

}  // namespace spamd
