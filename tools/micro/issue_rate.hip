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

KERNEL(k_or_vop2, "v_or_b32 v10, s43, v20\n\tv_or_b32 v11, s43, v20\n\tv_or_b32 v12, s43, v20\n\tv_or_b32 v13, s43, v20\n\t", "")
KERNEL(k_or_sdwa, "v_or_b32_sdwa v10, s43, v20 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\tv_or_b32_sdwa v11, s43, v20 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\tv_or_b32_sdwa v12, s43, v20 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\tv_or_b32_sdwa v13, s43, v20 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t", "")
KERNEL(k_andor_v, "v_and_or_b32 v10, v22, v21, v20\n\tv_and_or_b32 v11, v22, v21, v20\n\tv_and_or_b32 v12, v22, v21, v20\n\tv_and_or_b32 v13, v22, v21, v20\n\t", "")
KERNEL(k_fmac_s, "v_fmac_f32 v10, s43, v21\n\tv_fmac_f32 v11, s43, v21\n\tv_fmac_f32 v12, s43, v21\n\tv_fmac_f32 v13, s43, v21\n\t", "")
KERNEL(k_fmac_v, "v_fmac_f32 v10, v20, v21\n\tv_fmac_f32 v11, v20, v21\n\tv_fmac_f32 v12, v20, v21\n\tv_fmac_f32 v13, v20, v21\n\t", "")
KERNEL(k_pkfma_s, "v_pk_fma_f32 v[10:11], v[20:21], s[44:45], v[10:11] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\tv_pk_fma_f32 v[12:13], v[20:21], s[44:45], v[12:13] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\tv_pk_fma_f32 v[14:15], v[20:21], s[44:45], v[14:15] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\tv_pk_fma_f32 v[16:17], v[20:21], s[44:45], v[16:17] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t", "")
KERNEL(k_mov_s, "v_mov_b32 v10, s43\n\tv_mov_b32 v11, s43\n\tv_mov_b32 v12, s43\n\tv_mov_b32 v13, s43\n\t", "")
KERNEL(k_entry_pk, "v_and_or_b32 v26, s43, v21, v20\n\tds_read_b64 v[14:15], v26\n\ts_set_gpr_idx_idx s41\n\tv_pk_fma_f32 v[10:11], v[14:15], s[44:45], v[10:11] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t", "s_waitcnt lgkmcnt(0)\n\t")
KERNEL(k_entry_or_pk, "v_or_b32 v26, s43, v20\n\tds_read_b64 v[14:15], v26\n\ts_set_gpr_idx_idx s41\n\tv_pk_fma_f32 v[10:11], v[14:15], s[44:45], v[10:11] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t", "s_waitcnt lgkmcnt(0)\n\t")
KERNEL(k_entry_or_2fmac, "v_or_b32 v26, s43, v20\n\tds_read_b64 v[14:15], v26\n\ts_set_gpr_idx_idx s41\n\tv_fmac_f32 v10, s43, v14\n\tv_fmac_f32 v11, s43, v15\n\t", "s_waitcnt lgkmcnt(0)\n\t")

#define ENTRY_PK "v_and_or_b32 v26, s43, v21, v20\n\tds_read_b64 v[14:15], v26\n\ts_set_gpr_idx_idx s41\n\tv_pk_fma_f32 v[10:11], v[14:15], s[44:45], v[10:11] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
#define ENTRY8 ENTRY_PK ENTRY_PK ENTRY_PK ENTRY_PK ENTRY_PK ENTRY_PK ENTRY_PK ENTRY_PK
#define SCLOB CLOB, "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63"
// 8 entries + one scalar block load (scalar-cache hit: every wave re-reads its own 64-byte line)
__global__ void __launch_bounds__(1024) k_entry8_smem(int iters, const int* p, int mode) {
  extern __shared__ char lds[];
  const int* q = p + (blockIdx.x * 16 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * 16;
  asm volatile("v_mov_b32 v20, 0\n\tv_mov_b32 v21, 0\n\ts_mov_b32 s40, 2\n\ts_mov_b32 s41, 4\n\ts_mov_b32 s43, 0\n\ts_mov_b32 s44, 0\n\ts_mov_b32 s45, 0\n\t" ::: SCLOB);
  if (mode == 0)
    for (int i = 0; i < iters; ++i)
      asm volatile(".rept 8\n\t" ENTRY8 "s_waitcnt lgkmcnt(0)\n\ts_load_dwordx16 s[48:63], %0, 0x0\n\t.endr\n\ts_waitcnt lgkmcnt(0)\n\t" ::"s"(q) : SCLOB);
  else if (mode == 1)
    for (int i = 0; i < iters; ++i)
      asm volatile(".rept 8\n\t" ENTRY8 "s_waitcnt lgkmcnt(0)\n\t.endr\n\t" ::"s"(q) : SCLOB);
  else
    for (int i = 0; i < iters; ++i)
      asm volatile(".rept 8\n\t" ENTRY8 "s_waitcnt lgkmcnt(0)\n\ts_load_dwordx8 s[48:55], %0, 0x0\n\t.endr\n\ts_waitcnt lgkmcnt(0)\n\t" ::"s"(q) : SCLOB);
  if (p == nullptr) lds[threadIdx.x] = 1;
}

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

void run_smem(int mode, const char* name, int threads) {
  static int* d = nullptr;
  if (!d) { hipMalloc(&d, 1 << 24); hipMemset(d, 0, 1 << 24); }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 2000;
  k_entry8_smem<<<256, threads, 1024>>>(10, d, mode);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_entry8_smem<<<256, threads, 1024>>>(iters, d, mode);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double wps = threads / 256.0;
  printf("%-34s %2.0f waves/SIMD: %6.2f cycles per ENTRY per SIMD\n", name, wps, ms * 1e-3 * 2.4e9 / ((double)iters * 64 * wps));
}

int main() {
  for (int threads : {512, 1024}) {
    run_smem(1, "8 entries + wait", threads);
    run_smem(0, "8 entries + wait + s_load x16", threads);
    run_smem(2, "8 entries + wait + s_load x8", threads);
  }
  for (int threads : {512, 1024, 2048 / 2}) {
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
    run("or vop2 sgpr", k_or_vop2, 4, threads);
    run("or sdwa sgpr", k_or_sdwa, 4, threads);
    run("and_or vgpr", k_andor_v, 4, threads);
    run("fmac sgpr", k_fmac_s, 4, threads);
    run("fmac vgpr", k_fmac_v, 4, threads);
    run("pk_fma sgpr", k_pkfma_s, 4, threads);
    run("mov sgpr", k_mov_s, 4, threads);
    run("entry pk(4)", k_entry_pk, 4, threads);
    run("entry or+pk(4)", k_entry_or_pk, 4, threads);
    run("entry or+2fmac(5)", k_entry_or_2fmac, 5, threads);
    if (threads == 1024) break;
  }
  return 0;
}
