// Scalar-load latency probe: one wave per workgroup runs N dependent s_load_dwordx16 (the next address
// adds a loaded dword, which is always 0, to pointer + stride).  Prints ns per load for:
//   stride 0 (scalar-cache hit), 64 B over a small region (L2 hit after warm-up), 64 B over a huge region (HBM).
// hipcc --offload-arch=gfx950 -O3 tools/micro/smem_lat.hip -o /tmp/smem_lat && /tmp/smem_lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void chase(const int* p, long stride, long wrap, int n, int* sink) {
  const int* q = p + (long)blockIdx.x * 1024;  // separate start per workgroup
  long off = 0;
  int acc = 0;
  for (int i = 0; i < n; ++i) {
    int v;
    asm volatile("s_load_dwordx16 s[40:55], %1, 0x0\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, s47"
                 : "=s"(v)
                 : "s"(q + off)
                 : "memory", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52",
                   "s53", "s54", "s55");
    acc += v;
    off += stride / 4 + v;
    if (off >= wrap / 4) off = 0;
  }
  if (threadIdx.x == 0) sink[blockIdx.x] = acc;
}

int main() {
  const size_t bytes = 2ull << 30;
  int* d;
  int* sink;
  hipMalloc(&d, bytes);
  hipMemset(d, 0, bytes);
  hipMalloc(&sink, 4096 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  struct Case { const char* name; long stride, wrap; int blocks, threads; };
  std::vector<Case> cases = {
      {"K$ hit, 1 wave", 0, 1 << 20, 1, 64},
      {"L2 hit (64B stride, 256KB), 1 wave", 64, 256 << 10, 1, 64},
      {"L2 hit (64B stride, 2MB), 1 wave", 64, 2 << 20, 1, 64},
      {"HBM (64B stride, 1GB), 1 wave", 64, 1 << 30, 1, 64},
      {"HBM (4KB stride, 1GB), 1 wave", 4096, 1 << 30, 1, 64},
      {"K$ hit, 256 WG x 16 waves", 0, 1 << 20, 256, 1024},
      {"L2 hit (64B, 256KB), 256 WG x 16 waves", 64, 256 << 10, 256, 1024},
  };
  for (auto& c : cases) {
    const int n = 20000;
    chase<<<c.blocks, c.threads>>>(d, c.stride, c.wrap, n, sink);  // warm
    hipDeviceSynchronize();
    hipEventRecord(e0);
    chase<<<c.blocks, c.threads>>>(d, c.stride, c.wrap, n, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-45s %8.1f ns per load\n", c.name, ms * 1e6 / n);
  }
  return 0;
}
