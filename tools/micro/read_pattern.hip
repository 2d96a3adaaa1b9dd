// How fast can 0.8 GB be READ, by access pattern (MI355X)?   hipcc --offload-arch=gfx950 -O3 -o bin/read_pattern read_pattern.hip
//   A: every wave reads its own contiguous piece (4096 / 8192 / 16384 pieces), 16 B per lane, U loads in flight
//   B: grid-stride: wave w reads chunks w, w + W, ...  (the chip's reads form one moving window)
// two arrays (index + value) as the SpMV stream has them.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned u4 __attribute__((ext_vector_type(4)));

// C: pieces, but a lane reads TWO CONSECUTIVE vectors (32-byte lane stride: every load instruction touches 16 lines
// and uses half of each) - the stream kernel's "8 consecutive elements per lane"
__global__ void __launch_bounds__(1024) rd2(const u4* __restrict__ a, const u4* __restrict__ b, size_t nvec, unsigned* out) {
  const int lane = threadIdx.x & 63;
  const size_t W = (size_t)gridDim.x * (blockDim.x / 64), w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
  const size_t chunk = 128, nchunks = nvec / chunk;
  const size_t per = (nchunks + W - 1) / W;
  const size_t c0 = w * per, c1 = c0 + per < nchunks ? c0 + per : nchunks;
  unsigned acc = 0;
  for (size_t c = c0; c < c1; ++c) {
    const u4 x0 = a[c * chunk + 2 * lane], x1 = a[c * chunk + 2 * lane + 1];
    const u4 y0 = b[c * chunk + 2 * lane], y1 = b[c * chunk + 2 * lane + 1];
    acc += x0.x ^ x0.y ^ x0.z ^ x0.w ^ y0.x ^ y0.y ^ y0.z ^ y0.w ^ x1.x ^ x1.y ^ x1.z ^ x1.w ^ y1.x ^ y1.y ^ y1.z ^ y1.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int U, bool STRIDED>
__global__ void __launch_bounds__(1024) rd(const u4* __restrict__ a, const u4* __restrict__ b, size_t nvec, unsigned* out) {
  const int lane = threadIdx.x & 63;
  const size_t W = (size_t)gridDim.x * (blockDim.x / 64), w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
  const size_t chunk = 64 * U;                      // vectors per wave step
  const size_t nchunks = nvec / chunk;
  unsigned acc = 0;
  size_t c0, c1, cs;
  if (STRIDED) { c0 = w; c1 = nchunks; cs = W; }
  else { const size_t per = (nchunks + W - 1) / W; c0 = w * per; c1 = c0 + per < nchunks ? c0 + per : nchunks; cs = 1; }
  for (size_t c = c0; c < c1; c += cs) {
    u4 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = a[c * chunk + u * 64 + lane];
#pragma unroll
    for (int u = 0; u < U; ++u) y[u] = b[c * chunk + u * 64 + lane];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += x[u].x ^ x[u].y ^ x[u].z ^ x[u].w ^ y[u].x ^ y[u].y ^ y[u].z ^ y[u].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int U, bool S>
static void run(const char* name, const u4* a, const u4* b, size_t nvec, unsigned* out, int blocks, int threads) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((rd<U, S>), dim3(blocks), dim3(threads), 0, 0, a, b, nvec, out);
  hipEventRecord(e0);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((rd<U, S>), dim3(blocks), dim3(threads), 0, 0, a, b, nvec, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  printf("%-28s U=%d blocks=%5d x %4d: %.4f ms  %.2f TB/s\n", name, U, blocks, threads, ms, 2.0 * nvec * 16 / ms / 1e9);
}

int main() {
  const size_t nvec = 25000000;  // 400 MB per array
  u4 *a, *b; unsigned* out;
  hipMalloc(&a, nvec * 16); hipMalloc(&b, nvec * 16); hipMalloc(&out, 4);
  hipMemset(a, 1, nvec * 16); hipMemset(b, 2, nvec * 16);
  run<2, false>("pieces", a, b, nvec, out, 512, 512);
  run<2, false>("pieces", a, b, nvec, out, 1024, 512);
  run<2, false>("pieces", a, b, nvec, out, 256, 1024);
  run<2, false>("pieces", a, b, nvec, out, 2048, 256);
  run<4, false>("pieces", a, b, nvec, out, 512, 512);
  run<1, false>("pieces", a, b, nvec, out, 512, 512);
  run<1, false>("pieces", a, b, nvec, out, 1024, 512);
  for (int cfg = 0; cfg < 2; ++cfg) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = cfg ? 256 : 512, threads = cfg ? 1024 : 512;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(rd2, dim3(blocks), dim3(threads), 0, 0, a, b, nvec, out);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(rd2, dim3(blocks), dim3(threads), 0, 0, a, b, nvec, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("pieces, 32-byte lane stride   blocks=%5d x %4d: %.4f ms  %.2f TB/s\n", blocks, threads, ms, 2.0 * nvec * 16 / ms / 1e9);
  }
  run<2, true>("grid-stride", a, b, nvec, out, 512, 512);
  run<2, true>("grid-stride", a, b, nvec, out, 1024, 512);
  run<4, true>("grid-stride", a, b, nvec, out, 512, 512);
  run<1, true>("grid-stride", a, b, nvec, out, 512, 512);
  run<1, true>("grid-stride", a, b, nvec, out, 1024, 512);
  run<1, true>("grid-stride", a, b, nvec, out, 2048, 512);
  run<2, true>("grid-stride", a, b, nvec, out, 4096, 256);
  return 0;
}
