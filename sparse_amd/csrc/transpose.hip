// out = in^T for a dense row-major matrix: what `dense @ sparse` needs of its dense operand (the product runs as
// (sparse^T @ dense^T)^T, reference dispatch sparse/numba_backend/_common.py:339-503; `_dot.py` here).  The copy torch makes for
// `x.t().contiguous()` of a 128 x 10^6 float32 matrix runs at 1.3 TB/s (0.65-0.89 ms for 1 GB of traffic, a third of the
// product it prepares - tools/r06/fat_one.py, round 6); a 64 x 64 tile through LDS reads and writes whole 256-byte runs.
#include "common.h"

namespace spamd {

template <typename E>
__global__ void __launch_bounds__(256) transpose_tile_kernel(int64_t rows, int64_t cols, const E* __restrict__ in, int64_t ld_in,
                                                              E* __restrict__ out, int64_t ld_out, int64_t tiles_c) {
  __shared__ E tile[64][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t tr = (int64_t)blockIdx.x / tiles_c, tc = (int64_t)blockIdx.x % tiles_c;
  const int64_t r0 = tr * 64, c0 = tc * 64;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int64_t r = r0 + ty + 4 * j, c = c0 + tx;
    if (r < rows && c < cols) tile[ty + 4 * j][tx] = in[r * ld_in + c];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int64_t c = c0 + ty + 4 * j, r = r0 + tx;
    if (r < rows && c < cols) out[c * ld_out + r] = tile[tx][ty + 4 * j];
  }
}

}  // namespace spamd

// out[c * ld_out + r] = in[r * ld_in + c] for r < rows, c < cols; elements of 1, 2, 4 or 8 bytes moved bit-wise.
extern "C" int spamd_transpose_2d(int elem_bytes, int64_t rows, int64_t cols, const void* in, int64_t ld_in, void* out, int64_t ld_out,
                                  void* stream) {
  using namespace spamd;
  if (rows < 0 || cols < 0 || ld_in < cols || ld_out < rows) return SPAMD_EINVAL;
  if (rows == 0 || cols == 0) return 0;
  const int64_t tiles_r = ceil_div(rows, (int64_t)64), tiles_c = ceil_div(cols, (int64_t)64);
  if (tiles_r * tiles_c >= ((int64_t)1 << 31)) return SPAMD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)(tiles_r * tiles_c));
  switch (elem_bytes) {
    case 1: hipLaunchKernelGGL(transpose_tile_kernel<uint8_t>, grid, dim3(256), 0, s, rows, cols, (const uint8_t*)in, ld_in, (uint8_t*)out, ld_out, tiles_c); break;
    case 2: hipLaunchKernelGGL(transpose_tile_kernel<uint16_t>, grid, dim3(256), 0, s, rows, cols, (const uint16_t*)in, ld_in, (uint16_t*)out, ld_out, tiles_c); break;
    case 4: hipLaunchKernelGGL(transpose_tile_kernel<uint32_t>, grid, dim3(256), 0, s, rows, cols, (const uint32_t*)in, ld_in, (uint32_t*)out, ld_out, tiles_c); break;
    case 8: hipLaunchKernelGGL(transpose_tile_kernel<uint64_t>, grid, dim3(256), 0, s, rows, cols, (const uint64_t*)in, ld_in, (uint64_t*)out, ld_out, tiles_c); break;
    default: return SPAMD_ETYPE;
  }
  return launch_status();
}
