// rocPRIM's radix sort switches to a merge sort below `merge_sort_limit` items (default 2^20): time both algorithms for
// (int64 key, int64 payload) pairs at the sizes of BASELINE config 1 (10^6) and around.   hipcc -O3 --offload-arch=gfx950
#include <string.h>
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <vector>
#include <random>
template <class Config>
float run(size_t n, int bits, const int64_t* kin, int64_t* kout, const int64_t* vin, int64_t* vout) {
  size_t bytes = 0;
  rocprim::radix_sort_pairs<Config>(nullptr, bytes, kin, kout, vin, vout, n, 0, bits, 0);
  void* ws;
  hipMalloc(&ws, bytes);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) rocprim::radix_sort_pairs<Config>(ws, bytes, kin, kout, vin, vout, n, 0, bits, 0);
  hipEventRecord(a);
  for (int i = 0; i < 20; ++i) rocprim::radix_sort_pairs<Config>(ws, bytes, kin, kout, vin, vout, n, 0, bits, 0);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  hipFree(ws);
  return ms / 20;
}
int main() {
  using Merge = rocprim::default_config;
  using Sweep = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 32 * 1024>;
  for (size_t n : {100000ul, 300000ul, 1000000ul, 1048575ul, 1048577ul, 3000000ul}) {
    for (int bits : {20, 30, 40}) {
      std::vector<int64_t> h(n);
      std::mt19937_64 g(7);
      for (auto& x : h) x = (int64_t)(g() & ((1ull << bits) - 1));
      int64_t *kin, *kout, *vin, *vout;
      hipMalloc(&kin, n * 8); hipMalloc(&kout, n * 8); hipMalloc(&vin, n * 8); hipMalloc(&vout, n * 8);
      hipMemcpy(kin, h.data(), n * 8, hipMemcpyHostToDevice);
      hipMemcpy(vin, h.data(), n * 8, hipMemcpyHostToDevice);
      const float d = run<Merge>(n, bits, kin, kout, vin, vout);
      const float s = run<Sweep>(n, bits, kin, kout, vin, vout);
      printf("n=%zu bits=%d: default %.4f ms, merge_sort_limit=32K %.4f ms\n", n, bits, d, s);
      hipFree(kin); hipFree(kout); hipFree(vin); hipFree(vout);
    }
  }
  return 0;
}
