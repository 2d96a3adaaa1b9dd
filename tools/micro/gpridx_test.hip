// Micro-test: does s_set_gpr_idx_on (SRC2|DST) index a fixed accumulator block as expected on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k(float* out, const int* rows, int n) {
  const int lane = threadIdx.x & 63;
  asm volatile(
      ".set spamd_i, 128\n\t"
      ".rept 128\n\t"
      "v_mov_b32 v[spamd_i], 0\n\t"
      ".set spamd_i, spamd_i+1\n\t"
      ".endr\n\t" ::: "memory", "v128", "v255");
  for (int e = 0; e < n; ++e) {
    int r = __builtin_amdgcn_readfirstlane(rows[e]);
    float val = 2.0f + e, b = (float)lane;
    asm volatile(
        "s_lshl_b32 s40, %0, 1\n\t"
        "s_set_gpr_idx_on s40, gpr_idx(SRC2,DST)\n\t"
        "v_fma_f32 v128, %1, %2, v128\n\t"
        "v_fma_f32 v129, %1, %3, v129\n\t"
        "s_set_gpr_idx_off\n\t"
        : : "s"(r), "s"(val), "v"(b), "v"(b + 0.5f) : "s40", "m0", "scc", "memory", "v128", "v129", "v255");
  }
  // dump rows 0..3 (v128..v135)
  float o[8];
  asm volatile("v_mov_b32 %0, v128\n\tv_mov_b32 %1, v129\n\tv_mov_b32 %2, v130\n\tv_mov_b32 %3, v131\n\t"
               "v_mov_b32 %4, v132\n\tv_mov_b32 %5, v133\n\tv_mov_b32 %6, v134\n\tv_mov_b32 %7, v135"
               : "=v"(o[0]), "=v"(o[1]), "=v"(o[2]), "=v"(o[3]), "=v"(o[4]), "=v"(o[5]), "=v"(o[6]), "=v"(o[7]) : : "memory");
  for (int i = 0; i < 8; ++i) out[i * 64 + lane] = o[i];
}
int main() {
  float* out; int* rows; hipMalloc(&out, 8 * 64 * 4); hipMalloc(&rows, 16);
  int h[4] = {1, 3, 1, 0};
  hipMemcpy(rows, h, 16, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, rows, 4);
  float ho[512]; hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost);
  // expected at lane 5: row0: (5)*(5.0)=25, (5.5)*5=27.5 ; row1: 5*(2+4)=30, 5.5*6=33 ; row2: 0 ; row3: 5*3=15, 16.5
  for (int i = 0; i < 8; ++i) printf("acc[%d] lane5 = %g\n", i, ho[i * 64 + 5]);
  return 0;
}
