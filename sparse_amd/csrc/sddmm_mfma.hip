// A9, dense-tile form: SDDMM on the matrix cores for the 32 x 32 tiles of the mask that hold enough samples.
//
// The reference writes SDDMM as `s * (a @ b)` (examples/sddmm_example.py:51-52): a dense GEMM of the whole product,
// then a gather at the mask's coordinates.  The sampled kernel (sddmm.hip) forms only the sampled dot products and wins
// whenever the mask is thin everywhere (BASELINE config 4: 0.1 % uniform = one sample per tile).  A mask with populated
// blocks (block-diagonal / banded / clustered masks) is the other regime: a 32 x 32 tile with c samples costs the
// sampled kernel c row-pair gathers (2*K*2 B each) but the matrix core one 32 x 32 x K bf16 product (64 rows of K,
// K/16 v_mfma_f32_32x32x16_bf16) whatever c is.  The product dispatches PER TILE:
//
//   plan  (once per mask, cached on it): tile key = (row/32) * tile_cols + col/32 per sample; stable sort of the keys
//         with the sample index as payload; runs of equal keys = tiles; tiles with >= threshold samples go to the
//         matrix-core kernel, the samples of all other tiles to the sampled kernel.
//   tiles (this file): one wave per dense tile.  Lane l loads 16 bytes (8 bf16) of row l%32 of the A panel and of the Bt
//         panel at k = k0 + 8*(l/32) straight from global memory in the MFMA operand layout (both operands are
//         K-contiguous, so no LDS staging or transposition), K/16 MFMAs accumulate the tile in 16 VGPRs (fp32), the tile
//         goes to LDS (32 x 33 floats per wave) and the tile's samples pick their element: out[n] = s[n] * P[i][j].
//
// fp32 accumulate like the sampled kernel; the order of the K terms differs (the matrix core sums 16 products at a
// time), so the two paths agree to ~K * 2^-24 * sum|a_k b_k|, not bit for bit (tests/test_sddmm_gpu.py).
#include "common.h"
#include <hip/hip_bf16.h>

namespace spamd {

constexpr int SD_TILE = 32;

typedef __bf16 sd_bf16x8 __attribute__((ext_vector_type(8)));
typedef float sd_f32x16 __attribute__((ext_vector_type(16)));

template <typename I>
__global__ void __launch_bounds__(256) sddmm_tile_keys_kernel(int64_t nnz, const I* __restrict__ rows,
                                                              const I* __restrict__ cols, int64_t tile_cols,
                                                              int64_t* __restrict__ keys) {
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < nnz; n += (int64_t)gridDim.x * blockDim.x)
    keys[n] = ((int64_t)rows[n] / SD_TILE) * tile_cols + (int64_t)cols[n] / SD_TILE;
}

// seg_start[s] .. seg_start[s+1]: samples (positions in the tile-sorted order) of tile s.
// tile_flag[s] = tile s takes the matrix-core path; sample_flag[i] = sample at sorted position i is left to the sampled
// kernel.  Both are int64 flag arrays (n + 1 entries each, last one unused) for exclusive_scan + compact.
__global__ void __launch_bounds__(256) sddmm_classify_kernel(int64_t nseg, const int64_t* __restrict__ seg_start,
                                                             int64_t threshold, int64_t* __restrict__ tile_flag,
                                                             int64_t* __restrict__ sample_flag) {
  // one wave per tile: the tile's samples get their flag from the lanes
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t s = wave; s < nseg; s += nwaves) {
    const int64_t a = seg_start[s], b = seg_start[s + 1];
    const bool dense = b - a >= threshold;
    if (lane == 0) tile_flag[s] = dense ? 1 : 0;
    for (int64_t i = a + lane; i < b; i += 64) sample_flag[i] = dense ? 0 : 1;
  }
}

// One wave per dense tile.  tiles[d] = index of the tile's run in seg_start / position of its first sample.
template <typename I>
__global__ void __launch_bounds__(256) sddmm_mfma_kernel(int64_t ntiles, const int64_t* __restrict__ tiles,
                                                         const int64_t* __restrict__ seg_start,
                                                         const int64_t* __restrict__ keys_sorted,
                                                         const int64_t* __restrict__ perm, int64_t tile_cols, int64_t M,
                                                         int64_t N, const I* __restrict__ rows, const I* __restrict__ cols,
                                                         const float* __restrict__ s_data, const __bf16* __restrict__ A,
                                                         int64_t lda, const __bf16* __restrict__ Bt, int64_t ldb, int64_t K,
                                                         float* __restrict__ out) {
  __shared__ float tile_lds[4][SD_TILE][SD_TILE + 1];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int64_t d = (int64_t)blockIdx.x * 4 + wv;
  if (d >= ntiles) return;  // (whole waves leave; no barrier below: every wave owns its LDS slice)
  const int64_t s = tiles[d];
  const int64_t first = seg_start[s], last = seg_start[s + 1];
  const int64_t key = keys_sorted[first];
  const int64_t tr = key / tile_cols, tc = key - tr * tile_cols;
  int64_t ar = tr * SD_TILE + (lane & 31), bc = tc * SD_TILE + (lane & 31);
  if (ar >= M) ar = M - 1;  // edge tiles: clamped rows feed elements no sample refers to
  if (bc >= N) bc = N - 1;
  const __bf16* ap = A + ar * lda + (lane >> 5) * 8;
  const __bf16* bp = Bt + bc * ldb + (lane >> 5) * 8;
  sd_f32x16 acc;
#pragma unroll
  for (int v = 0; v < 16; ++v) acc[v] = 0.f;
  // K is a multiple of 16 (checked by the caller); four k-steps of loads in flight before their MFMAs
  int64_t k = 0;
  for (; k + 64 <= K; k += 64) {
    sd_bf16x8 a0 = *reinterpret_cast<const sd_bf16x8*>(ap + k), b0 = *reinterpret_cast<const sd_bf16x8*>(bp + k);
    sd_bf16x8 a1 = *reinterpret_cast<const sd_bf16x8*>(ap + k + 16), b1 = *reinterpret_cast<const sd_bf16x8*>(bp + k + 16);
    sd_bf16x8 a2 = *reinterpret_cast<const sd_bf16x8*>(ap + k + 32), b2 = *reinterpret_cast<const sd_bf16x8*>(bp + k + 32);
    sd_bf16x8 a3 = *reinterpret_cast<const sd_bf16x8*>(ap + k + 48), b3 = *reinterpret_cast<const sd_bf16x8*>(bp + k + 48);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b3, acc, 0, 0, 0);
  }
  for (; k < K; k += 16) {
    sd_bf16x8 a0 = *reinterpret_cast<const sd_bf16x8*>(ap + k), b0 = *reinterpret_cast<const sd_bf16x8*>(bp + k);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc, 0, 0, 0);
  }
  // D[i][j]: j = lane % 32, i = (v % 4) + 8 * (v / 4) + 4 * (lane / 32)
  float(*P)[SD_TILE + 1] = tile_lds[wv];
#pragma unroll
  for (int v = 0; v < 16; ++v) P[(v & 3) + 8 * (v >> 2) + 4 * (lane >> 5)][lane & 31] = acc[v];
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes have landed
  const int64_t r0 = tr * SD_TILE, c0 = tc * SD_TILE;
  for (int64_t i = first + lane; i < last; i += 64) {
    const int64_t n = perm[i];
    const int li = (int)((int64_t)rows[n] - r0), lj = (int)((int64_t)cols[n] - c0);
    out[n] = s_data[n] * P[li][lj];
  }
}

// K = 16 * KS known at compile time (64, 128, 256): a wave takes SD_GROUP consecutive dense tiles of the list (sorted by
// tile row, then tile column) and keeps the A panel of the current tile row in registers (KS operand registers of 16 bytes
// per lane): tiles of one tile row - the common case for banded / block-diagonal masks - then fetch only their Bt panel,
// i.e. 16 KB instead of 32 KB per tile at K = 256 (the tile kernel is bound by fetching its panels, not by the MFMAs).
constexpr int SD_GROUP = 4;

template <typename I, int KS>
__global__ void __launch_bounds__(256) sddmm_mfma_rowreuse_kernel(int64_t ntiles, const int64_t* __restrict__ tiles,
                                                                  const int64_t* __restrict__ seg_start,
                                                                  const int64_t* __restrict__ keys_sorted,
                                                                  const int64_t* __restrict__ perm, int64_t tile_cols, int64_t M,
                                                                  int64_t N, const I* __restrict__ rows,
                                                                  const I* __restrict__ cols, const float* __restrict__ s_data,
                                                                  const __bf16* __restrict__ A, int64_t lda,
                                                                  const __bf16* __restrict__ Bt, int64_t ldb,
                                                                  float* __restrict__ out) {
  __shared__ float tile_lds[4][SD_TILE][SD_TILE + 1];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  float(*P)[SD_TILE + 1] = tile_lds[wv];
  const int64_t d0 = ((int64_t)blockIdx.x * 4 + wv) * SD_GROUP;
  sd_bf16x8 areg[KS];
  int64_t cur_tr = -1;
  for (int g = 0; g < SD_GROUP; ++g) {
    const int64_t d = d0 + g;
    if (d >= ntiles) break;
    const int64_t s = tiles[d];
    const int64_t first = seg_start[s], last = seg_start[s + 1];
    const int64_t key = keys_sorted[first];
    const int64_t tr = key / tile_cols, tc = key - tr * tile_cols;
    if (tr != cur_tr) {
      int64_t ar = tr * SD_TILE + (lane & 31);
      if (ar >= M) ar = M - 1;
      const __bf16* ap = A + ar * lda + (lane >> 5) * 8;
#pragma unroll
      for (int q = 0; q < KS; ++q) areg[q] = *reinterpret_cast<const sd_bf16x8*>(ap + q * 16);
      cur_tr = tr;
    }
    int64_t bc = tc * SD_TILE + (lane & 31);
    if (bc >= N) bc = N - 1;
    const __bf16* bp = Bt + bc * ldb + (lane >> 5) * 8;
    sd_f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll
    for (int q0 = 0; q0 < KS; q0 += 4) {
      sd_bf16x8 b0 = *reinterpret_cast<const sd_bf16x8*>(bp + q0 * 16), b1 = *reinterpret_cast<const sd_bf16x8*>(bp + q0 * 16 + 16);
      sd_bf16x8 b2 = *reinterpret_cast<const sd_bf16x8*>(bp + q0 * 16 + 32), b3 = *reinterpret_cast<const sd_bf16x8*>(bp + q0 * 16 + 48);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(areg[q0], b0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(areg[q0 + 1], b1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(areg[q0 + 2], b2, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(areg[q0 + 3], b3, acc, 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < 16; ++v) P[(v & 3) + 8 * (v >> 2) + 4 * (lane >> 5)][lane & 31] = acc[v];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    const int64_t r0 = tr * SD_TILE, c0 = tc * SD_TILE;
    for (int64_t i = first + lane; i < last; i += 64) {
      const int64_t n = perm[i];
      const int li = (int)((int64_t)rows[n] - r0), lj = (int)((int64_t)cols[n] - c0);
      out[n] = s_data[n] * P[li][lj];
    }
    __builtin_amdgcn_wave_barrier();   // the next tile overwrites P: this wave's LDS reads above are issued before it
  }
}

}  // namespace spamd

using namespace spamd;

static unsigned sd_blocks(int64_t n, int per_block) {
  int64_t b = ceil_div(n, (int64_t)per_block);
  if (b > 65535 * 8) b = 65535 * 8;
  if (b < 1) b = 1;
  return (unsigned)b;
}

extern "C" int spamd_sddmm_tile_size(void) { return SD_TILE; }

extern "C" int spamd_sddmm_tile_keys(int idx_dtype, int64_t nnz, const void* rows, const void* cols, int64_t tile_cols,
                                     int64_t* keys, void* stream) {
  if (nnz < 0 || tile_cols <= 0) return SPAMD_EINVAL;
  if (nnz == 0) return 0;
  SPAMD_DISPATCH_IDX(idx_dtype, I, hipLaunchKernelGGL(sddmm_tile_keys_kernel<I>, dim3(sd_blocks(nnz, 256)), dim3(256), 0,
                                                      (hipStream_t)stream, nnz, (const I*)rows, (const I*)cols, tile_cols, keys))
  return launch_status();
}

extern "C" int spamd_sddmm_tile_classify(int64_t nseg, const int64_t* seg_start, int64_t threshold, int64_t* tile_flag,
                                         int64_t* sample_flag, void* stream) {
  if (nseg < 0 || threshold < 1) return SPAMD_EINVAL;
  if (nseg == 0) return 0;
  hipLaunchKernelGGL(sddmm_classify_kernel, dim3(sd_blocks(nseg, 4)), dim3(256), 0, (hipStream_t)stream, nseg, seg_start,
                     threshold, tile_flag, sample_flag);
  return launch_status();
}

extern "C" int spamd_sddmm_mfma_tiles(int idx_dtype, int64_t ntiles, const int64_t* tiles, const int64_t* seg_start,
                                      const int64_t* keys_sorted, const int64_t* perm, int64_t tile_cols, int64_t M, int64_t N,
                                      const void* rows, const void* cols, const float* s_data, const void* A, int64_t lda,
                                      const void* Bt, int64_t ldb, int64_t K, float* out, void* stream) {
  if (ntiles < 0 || K <= 0 || K % 16 != 0 || M <= 0 || N <= 0 || tile_cols <= 0) return SPAMD_EINVAL;
  if (ntiles == 0) return 0;
  if (((uintptr_t)A % 16) || ((uintptr_t)Bt % 16) || ((lda * 2) % 16) || ((ldb * 2) % 16)) return SPAMD_EINVAL;
#define SD_ROWREUSE(KS)                                                                                                    \
  SPAMD_DISPATCH_IDX(idx_dtype, I,                                                                                          \
                     hipLaunchKernelGGL((sddmm_mfma_rowreuse_kernel<I, KS>), dim3((unsigned)ceil_div(ntiles, (int64_t)(4 * SD_GROUP))), \
                                        dim3(256), 0, (hipStream_t)stream, ntiles, tiles, seg_start, keys_sorted, perm, tile_cols, \
                                        M, N, (const I*)rows, (const I*)cols, s_data, (const __bf16*)A, lda, (const __bf16*)Bt,  \
                                        ldb, out))                                                                          \
  return launch_status();
  if (ntiles >= 4 * SD_GROUP * 1024) {   // (a short tile list fills the chip better with one wave per tile)
    if (K == 256) { SD_ROWREUSE(16) }
    if (K == 128) { SD_ROWREUSE(8) }
    if (K == 64) { SD_ROWREUSE(4) }
  }
#undef SD_ROWREUSE
  SPAMD_DISPATCH_IDX(idx_dtype, I,
                     hipLaunchKernelGGL(sddmm_mfma_kernel<I>, dim3((unsigned)ceil_div(ntiles, (int64_t)4)), dim3(256), 0,
                                        (hipStream_t)stream, ntiles, tiles, seg_start, keys_sorted, perm, tile_cols, M, N,
                                        (const I*)rows, (const I*)cols, s_data, (const __bf16*)A, lda, (const __bf16*)Bt, ldb,
                                        K, out))
  return launch_status();
}
