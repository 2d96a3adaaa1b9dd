// Instruction-issue probe for gfx950: cycles per instruction per SIMD for the instruction mixes the
// tiled SpMM inner loop is made of, at 1 and 4 waves per SIMD.  Each kernel runs ITER x 64 copies of a
// pattern; time * clock / (ITER * 64 * instructions_in_pattern * waves_per_SIMD) is printed.
// hipcc --offload-arch=gfx950 -O3 tools/micro/issue_rate.hip -o issue_rate && ./issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define CLOB "memory", "scc", "m0", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "v10", "v11", "v12", "v13", "v14", "v15", \
             "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33"
#define KERNEL(name, body, tail)                                                   \
  __global__ void __launch_bounds__(1024) name(int iters, float* sink) {          \
    extern __shared__ char lds[];                                                  \
    asm volatile("v_mov_b32 v20, 0\n\tv_mov_b32 v21, 0\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0\n\t" \
                 "s_mov_b32 s40, 2\n\ts_mov_b32 s41, 4\n\ts_mov_b32 s42, 6\n\ts_mov_b32 s43, 0x3f800000\n\t" ::: CLOB); \
    for (int i = 0; i < iters; ++i) asm volatile(".rept 64\n\t" body ".endr\n\t" tail ::: CLOB); \
    if (sink == nullptr) lds[threadIdx.x] = 1;                                     \
  }

KERNEL(k_fma, "v_fma_f32 v10, v20, v21, v10\n\tv_fma_f32 v11, v20, v21, v11\n\tv_fma_f32 v12, v20, v21, v12\n\tv_fma_f32 v13, v20, v21, v13\n\t", "")
KERNEL(k_pkfma, "v_pk_fma_f32 v[10:11], v[20:21], v[22:23], v[10:11]\n\tv_pk_fma_f32 v[12:13], v[20:21], v[22:23], v[12:13]\n\tv_pk_fma_f32 v[14:15], v[20:21], v[22:23], v[14:15]\n\tv_pk_fma_f32 v[16:17], v[20:21], v[22:23], v[16:17]\n\t", "")
KERNEL(k_fma_sgpr, "v_fma_f32 v10, s43, v21, v10\n\tv_fma_f32 v11, s43, v21, v11\n\tv_fma_f32 v12, s43, v21, v12\n\tv_fma_f32 v13, s43, v21, v13\n\t", "")
KERNEL(k_salu, "s_add_u32 s44, s44, 1\n\ts_add_u32 s45, s45, 1\n\ts_add_u32 s46, s46, 1\n\ts_add_u32 s47, s47, 1\n\t", "")
KERNEL(k_fma_salu, "v_fma_f32 v10, v20, v21, v10\n\ts_add_u32 s44, s44, 1\n\tv_fma_f32 v11, v20, v21, v11\n\ts_add_u32 s45, s45, 1\n\t", "")
KERNEL(k_andor, "v_and_or_b32 v10, s43, v21, v20\n\tv_and_or_b32 v11, s43, v21, v20\n\tv_and_or_b32 v12, s43, v21, v20\n\tv_and_or_b32 v13, s43, v21, v20\n\t", "")
KERNEL(k_dsread, "ds_read_b64 v[10:11], v20\n\tds_read_b64 v[12:13], v20\n\tds_read_b64 v[14:15], v20\n\tds_read_b64 v[16:17], v20\n\t", "s_waitcnt lgkmcnt(0)\n\t")
KERNEL(k_ds_fma, "ds_read_b64 v[10:11], v20\n\tv_fma_f32 v24, v20, v21, v24\n\tds_read_b64 v[12:13], v20\n\tv_fma_f32 v25, v20, v21, v25\n\t", "s_waitcnt lgkmcnt(0)\n\t")
KERNEL(k_idx_fma, "s_set_gpr_idx_on s40, gpr_idx(SRC2,DST)\n\tv_fma_f32 v10, s43, v21, v10\n\ts_set_gpr_idx_idx s41\n\tv_fma_f32 v10, s43, v21, v10\n\ts_set_gpr_idx_idx s42\n\tv_fma_f32 v10, s43, v21, v10\n\ts_set_gpr_idx_off\n\t", "")
KERNEL(k_idx_only, "s_set_gpr_idx_on s40, gpr_idx(SRC2,DST)\n\ts_set_gpr_idx_idx s41\n\ts_set_gpr_idx_idx s42\n\ts_set_gpr_idx_off\n\t", "")
KERNEL(k_idxmode_fma, "v_fma_f32 v10, s43, v21, v10\n\tv_fma_f32 v11, s43, v21, v11\n\tv_fma_f32 v12, s43, v21, v12\n\tv_fma_f32 v13, s43, v21, v13\n\t", "")
// the inner loop of the tiled kernel for one entry, without scalar loads: and_or, ds_read, idx, fma, fma
KERNEL(k_entry, "v_and_or_b32 v26, s43, v21, v20\n\tds_read_b64 v[14:15], v26\n\ts_set_gpr_idx_idx s41\n\tv_fma_f32 v10, s43, v14, v10\n\tv_fma_f32 v11, s43, v15, v11\n\t", "s_waitcnt lgkmcnt(0)\n\t")

template <typename K>
void run(const char* name, K kern, int ninstr, int threads) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 2000;
  kern<<<256, threads, 1024>>>(10, (float*)1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<<<256, threads, 1024>>>(iters, (float*)1);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double wps = threads / 256.0;
  const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 64 * ninstr * wps);
  printf("%-14s %2.0f waves/SIMD: %6.2f cycles per instruction per SIMD (at 2.4 GHz)\n", name, wps, cyc);
}

int main() {
  for (int threads : {256, 1024, 2048 / 2}) {
    run("fma", k_fma, 4, threads);
    run("pk_fma", k_pkfma, 4, threads);
    run("fma sgpr", k_fma_sgpr, 4, threads);
    run("salu", k_salu, 4, threads);
    run("fma+salu", k_fma_salu, 4, threads);
    run("and_or", k_andor, 4, threads);
    run("ds_read_b64", k_dsread, 4, threads);
    run("ds+fma", k_ds_fma, 4, threads);
    run("idx+fma", k_idx_fma, 7, threads);
    run("idx only", k_idx_only, 4, threads);
    run("entry(5)", k_entry, 5, threads);
    if (threads == 1024) break;
  }
  return 0;
}
