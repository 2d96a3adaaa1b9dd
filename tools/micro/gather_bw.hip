// Micro-benchmark: how fast can the chip gather random 512-byte rows of a K x 128 fp32 table?
// (the inner operation of CSR x dense SpMM at N=128).  Indices come from an LCG so there is no
// index stream; each wave keeps U independent gathers in flight.  Variants:
//   mode 0: 64 lanes x 8 B  (one row per wave-instruction, global_load_dwordx2)
//   mode 1: 2 x 32 lanes x 16 B (two rows per wave-instruction, global_load_dwordx4)
//   mode 2: LDS-DMA  global_load_lds_dwordx4, two rows per instruction, read back with ds_read_b64
// Build: hipcc --offload-arch=gfx950 -O3 gather_bw.hip -o gather_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int U>
__global__ void __launch_bounds__(256) gather_x2(const float* __restrict__ tab, unsigned K, int iters, float* out) {
  const int lane = threadIdx.x & 63;
  unsigned seed = __builtin_amdgcn_readfirstlane((blockIdx.x * 4 + (threadIdx.x >> 6)) * 2654435761u + 12345u);
  float2 acc = {0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    float2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      unsigned r = lcg(seed) % K;
      v[u] = *reinterpret_cast<const float2*>(tab + (size_t)r * 128 + lane * 2);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; }
  }
  if (acc.x == 1.2345f) out[threadIdx.x] = acc.x + acc.y;
}

template <int U>
__global__ void __launch_bounds__(256) gather_x4(const float* __restrict__ tab, unsigned K, int iters, float* out) {
  const int lane = threadIdx.x & 63;
  unsigned seed = __builtin_amdgcn_readfirstlane((blockIdx.x * 4 + (threadIdx.x >> 6)) * 2654435761u + 12345u);
  float4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      unsigned r0 = lcg(seed) % K, r1 = lcg(seed) % K;
      unsigned r = lane < 32 ? r0 : r1;
      v[u] = *reinterpret_cast<const float4*>(tab + (size_t)r * 128 + (lane & 31) * 4);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  if (acc.x == 1.2345f) out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

// LDS-DMA ring: each wave owns DEPTH slots of 1 KiB; steady state keeps DEPTH-1 DMAs in flight.
template <int DEPTH>
__global__ void __launch_bounds__(256) gather_lds(const float* __restrict__ tab, unsigned K, int iters, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* ring = smem + wv * DEPTH * 1024;
  unsigned seed = __builtin_amdgcn_readfirstlane((blockIdx.x * 4 + wv) * 2654435761u + 12345u);
  float2 acc = {0.f, 0.f};
  auto issue = [&](int slot) {
    unsigned r0 = lcg(seed) % K, r1 = lcg(seed) % K;
    unsigned r = lane < 32 ? r0 : r1;
    const float* src = tab + (size_t)r * 128 + (lane & 31) * 4;
    unsigned ldsaddr = (unsigned)(size_t)(ring + slot * 1024);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(ldsaddr), "v"(src) : "memory", "m0");
  };
#pragma unroll
  for (int s = 0; s < DEPTH; ++s) issue(s);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) {
      asm volatile("s_waitcnt vmcnt(%0)" : : "n"(DEPTH - 1) : "memory");
      const float2 a = *reinterpret_cast<const float2*>(ring + s * 1024 + lane * 8);
      const float2 b = *reinterpret_cast<const float2*>(ring + s * 1024 + 512 + lane * 8);
      acc.x += a.x + b.x; acc.y += a.y + b.y;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      issue(s);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc.x == 1.2345f) out[threadIdx.x] = acc.x + acc.y;
}

template <typename F>
double time_it(F launch, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main(int argc, char** argv) {
  const int blocks_per_cu = argc > 1 ? atoi(argv[1]) : 8;
  float* out; CK(hipMalloc(&out, 4096));
  for (unsigned K : {1000u, 5000u, 10000u, 100000u, 1000000u}) {
    float* tab; CK(hipMalloc(&tab, (size_t)K * 512)); CK(hipMemset(tab, 0, (size_t)K * 512));
    const int grid = 256 * blocks_per_cu, iters = 400;
    auto report = [&](const char* name, double ms, double rows_per_wave) {
      double bytes = (double)grid * 4 * rows_per_wave * 512.0;
      printf("K=%7u (%6.1f MB) %-22s %8.3f ms  %7.2f TB/s  (%.1f B/clk/CU @2.4GHz)\n", K, K * 512.0 / 1e6, name, ms,
             bytes / ms / 1e9, bytes / ms / 1e-3 / 256 / 2.4e9);
    };
    report("x2 U=4", time_it([&] { hipLaunchKernelGGL(gather_x2<4>, dim3(grid), dim3(256), 0, 0, tab, K, iters, out); }, 5), iters * 4.0);
    report("x2 U=8", time_it([&] { hipLaunchKernelGGL(gather_x2<8>, dim3(grid), dim3(256), 0, 0, tab, K, iters, out); }, 5), iters * 8.0);
    report("x2 U=16", time_it([&] { hipLaunchKernelGGL(gather_x2<16>, dim3(grid), dim3(256), 0, 0, tab, K, iters, out); }, 5), iters * 16.0);
    report("x4(2rows) U=4", time_it([&] { hipLaunchKernelGGL(gather_x4<4>, dim3(grid), dim3(256), 0, 0, tab, K, iters, out); }, 5), iters * 8.0);
    report("x4(2rows) U=8", time_it([&] { hipLaunchKernelGGL(gather_x4<8>, dim3(grid), dim3(256), 0, 0, tab, K, iters, out); }, 5), iters * 16.0);
    report("ldsdma D=4", time_it([&] { hipLaunchKernelGGL(gather_lds<4>, dim3(grid), dim3(256), 4 * 4 * 1024, 0, tab, K, iters, out); }, 5), (iters + 1) * 8.0);
    report("ldsdma D=8", time_it([&] { hipLaunchKernelGGL(gather_lds<8>, dim3(grid), dim3(256), 4 * 8 * 1024, 0, tab, K, iters / 2, out); }, 5), (iters / 2 + 1) * 16.0);
    CK(hipFree(tab));
  }
  return 0;
}
