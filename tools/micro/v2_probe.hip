// Probe for the second-generation tiled-SpMM entry pipeline: per stored element
//   * the LDS address comes from a VGPR that holds 16 entries' row offsets (one per lane of a 16-lane DPP row,
//     replicated over the four rows) through `v_add_u32_dpp ... row_newbcast:n` (no SGPR operand),
//   * the multiplicand comes from a VGPR that holds 16 values the same way, consumed by `v_fmac_f32_dpp`,
//   * the accumulator index comes from a 16-bit scalar stream word (0x8000 | index: DST_REL + M0[7:0]) moved into M0
//     with one SALU instruction.
// Part 1 checks the semantics on the device (row_newbcast, VOP2 fmac under DST_REL with raw M0 writes); part 2
// measures cycles per entry per SIMD of the old and the new instruction mix in a two-data-set software pipeline.
// hipcc --offload-arch=gfx950 -O3 tools/micro/v2_probe.hip -o v2_probe && ./v2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

// ---------------------------------------------------------------- part 1: semantics
// entries e = 0..7 go to rows rows[e]; value of entry e = 2 + e; B row = (lane + 0.25, lane + 0.75)
__global__ void __launch_bounds__(64) sem_kernel(float* out, const int* rows, int mode_bits) {
  const int lane = threadIdx.x & 63;
  float vals = 2.0f + (float)(lane & 15);
  float b0 = lane + 0.25f, b1 = lane + 0.75f;
  asm volatile(
      ".set spamd_i, 128\n\t"
      ".rept 16\n\t"
      "v_mov_b32 v[spamd_i], 0\n\t"
      ".set spamd_i, spamd_i+1\n\t"
      ".endr\n\t" ::: "memory", "v128", "v143");
  int m[8];
  for (int e = 0; e < 8; ++e) m[e] = __builtin_amdgcn_readfirstlane((mode_bits << 12) | (2 * rows[e]));
  // packed halves: (m1 << 16 | m0) ...
  int p0 = m[0] | (m[1] << 16), p1 = m[2] | (m[3] << 16), p2 = m[4] | (m[5] << 16), p3 = m[6] | (m[7] << 16);
#define ENT(n, setm0)                                                                     \
  setm0 "\n\t"                                                                            \
  "v_fmac_f32_dpp v128, %4, %5 row_newbcast:" #n " row_mask:0xf bank_mask:0xf\n\t"      \
  "v_fmac_f32_dpp v129, %4, %6 row_newbcast:" #n " row_mask:0xf bank_mask:0xf\n\t"
  asm volatile(
      "s_set_gpr_idx_on %0, gpr_idx(DST)\n\t"
      ENT(0, "s_mov_b32 m0, %0") ENT(1, "s_lshr_b32 m0, %0, 16")
      ENT(2, "s_mov_b32 m0, %1") ENT(3, "s_lshr_b32 m0, %1, 16")
      ENT(4, "s_mov_b32 m0, %2") ENT(5, "s_lshr_b32 m0, %2, 16")
      ENT(6, "s_mov_b32 m0, %3") ENT(7, "s_lshr_b32 m0, %3, 16")
      "s_set_gpr_idx_off\n\t"
      :
      : "s"(p0), "s"(p1), "s"(p2), "s"(p3), "v"(vals), "v"(b0), "v"(b1)
      : "m0", "scc", "memory", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v143");
  float o[8];
  asm volatile("v_mov_b32 %0, v128\n\tv_mov_b32 %1, v129\n\tv_mov_b32 %2, v130\n\tv_mov_b32 %3, v131\n\t"
               "v_mov_b32 %4, v132\n\tv_mov_b32 %5, v133\n\tv_mov_b32 %6, v134\n\tv_mov_b32 %7, v135"
               : "=v"(o[0]), "=v"(o[1]), "=v"(o[2]), "=v"(o[3]), "=v"(o[4]), "=v"(o[5]), "=v"(o[6]), "=v"(o[7]) : : "memory");
  for (int i = 0; i < 8; ++i) out[i * 64 + lane] = o[i];
}

// address path: v_add_u32_dpp into the data register, ds_read_b64 from it
__global__ void __launch_bounds__(64) addr_kernel(float* out) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  for (int i = lane; i < 16 * 128; i += 64) lds[i] = (float)i;
  __syncthreads();
  int offs = (lane & 15) * 512;  // row offset of entry n = lane & 15
  int base = lane * 8;
  float r0, r1;
  asm volatile(
      "s_nop 4\n\t"
      "v_add_u32_dpp v130, %2, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
      "ds_read_b64 v[130:131], v130\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_mov_b32 %0, v130\n\tv_mov_b32 %1, v131\n\t"
      : "=v"(r0), "=v"(r1)
      : "v"(offs), "v"(base)
      : "memory", "v130", "v131");
  out[lane * 2] = r0;
  out[lane * 2 + 1] = r1;
}

// ---------------------------------------------------------------- part 2: rates
#define CLOBV "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", \
              "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", \
              "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v127"
#define CLOBS "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55"
#define CLOB "memory", "scc", "m0", CLOBV, CLOBS

// old mix: d0 in s(40+2i), value in s(41+2i)
#define O_P1(i, d) "v_and_or_b32 v4" #i ", s40, v61, v60\n\t"
#define OLD_P1(D)                                                                                         \
  "v_and_or_b32 v40, s40, v61, v60\n\tv_and_or_b32 v41, s42, v61, v60\n\tv_and_or_b32 v42, s44, v61, v60\n\tv_and_or_b32 v43, s46, v61, v60\n\t" \
  "ds_read_b64 v[" D "+0:" D "+1], v40\n\tds_read_b64 v[" D "+2:" D "+3], v41\n\tds_read_b64 v[" D "+4:" D "+5], v42\n\tds_read_b64 v[" D "+6:" D "+7], v43\n\t" \
  "v_and_or_b32 v40, s48, v61, v60\n\tv_and_or_b32 v41, s50, v61, v60\n\tv_and_or_b32 v42, s52, v61, v60\n\tv_and_or_b32 v43, s54, v61, v60\n\t" \
  "ds_read_b64 v[" D "+8:" D "+9], v40\n\tds_read_b64 v[" D "+10:" D "+11], v41\n\tds_read_b64 v[" D "+12:" D "+13], v42\n\tds_read_b64 v[" D "+14:" D "+15], v43\n\t"
#define OLD_FMA(D, k, s0, s1) \
  "s_set_gpr_idx_idx s" #s0 "\n\tv_pk_fma_f32 v[62:63], v[" D "+" #k ":" D "+" #k "+1], s[" #s0 ":" #s1 "], v[62:63] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
#define OLD_P2(D)                                                                                         \
  "s_set_gpr_idx_on s40, gpr_idx(SRC2,DST)\n\t"                                                           \
  OLD_FMA(D, 0, 40, 41) OLD_FMA(D, 2, 42, 43) OLD_FMA(D, 4, 44, 45) OLD_FMA(D, 6, 46, 47)                 \
  OLD_FMA(D, 8, 48, 49) OLD_FMA(D, 10, 50, 51) OLD_FMA(D, 12, 52, 53) OLD_FMA(D, 14, 54, 55)              \
  "s_set_gpr_idx_off\n\t"

// new mix: offsets of 16 entries in v58, values in v59 (lane n of every DPP row), packed M0 words in s40..s43
#define NA(D, k, n) "v_add_u32_dpp v[" D "+" #k "], v58, v60 row_newbcast:" #n " row_mask:0xf bank_mask:0xf\n\t"
#define NR(D, k) "ds_read_b64 v[" D "+" #k ":" D "+" #k "+1], v[" D "+" #k "]\n\t"
#define NEW_P1(D, n0, n1, n2, n3, n4, n5, n6, n7)                                                          \
  NA(D, 0, n0) NA(D, 2, n1) NA(D, 4, n2) NA(D, 6, n3) NR(D, 0) NR(D, 2) NR(D, 4) NR(D, 6)                    \
  NA(D, 8, n4) NA(D, 10, n5) NA(D, 12, n6) NA(D, 14, n7) NR(D, 8) NR(D, 10) NR(D, 12) NR(D, 14)
#define NF(D, k, n, setm0)                                                                                 \
  setm0 "\n\t"                                                                                             \
  "v_fmac_f32_dpp v62, v59, v[" D "+" #k "] row_newbcast:" #n " row_mask:0xf bank_mask:0xf\n\t"           \
  "v_fmac_f32_dpp v63, v59, v[" D "+" #k "+1] row_newbcast:" #n " row_mask:0xf bank_mask:0xf\n\t"
#define NEW_P2(D, n0, n1, n2, n3, n4, n5, n6, n7, sa, sb, sc, sd)                                           \
  "s_set_gpr_idx_on s" #sa ", gpr_idx(DST)\n\t"                                                           \
  NF(D, 0, n0, "s_mov_b32 m0, s" #sa) NF(D, 2, n1, "s_lshr_b32 m0, s" #sa ", 16")                          \
  NF(D, 4, n2, "s_mov_b32 m0, s" #sb) NF(D, 6, n3, "s_lshr_b32 m0, s" #sb ", 16")                          \
  NF(D, 8, n4, "s_mov_b32 m0, s" #sc) NF(D, 10, n5, "s_lshr_b32 m0, s" #sc ", 16")                         \
  NF(D, 12, n6, "s_mov_b32 m0, s" #sd) NF(D, 14, n7, "s_lshr_b32 m0, s" #sd ", 16")                        \
  "s_set_gpr_idx_off\n\t"
// new mix with a packed fma: the value is first broadcast into a register pair's low half by v_mov_b32_dpp
#define NM(T, n) "v_mov_b32_dpp v[" T "], v59 row_newbcast:" #n " row_mask:0xf bank_mask:0xf\n\t"
#define NPK(D, k, T, setm0) setm0 "\n\tv_pk_fma_f32 v[62:63], v[" D "+" #k ":" D "+" #k "+1], v[" T ":" T "+1], v[62:63] op_sel_hi:[1,0,1]\n\t"

#define PROLOGUE                                                                                            \
  "v_mov_b32 v60, %0\n\tv_mov_b32 v61, 0xfffffe00\n\tv_mov_b32 v58, 0\n\tv_mov_b32 v59, 0\n\t"             \
  "s_mov_b32 s40, 0x80028002\n\ts_mov_b32 s41, 0x80048004\n\ts_mov_b32 s42, 0x80068006\n\ts_mov_b32 s43, 0x80088008\n\t" \
  "s_mov_b32 s44, 0\n\ts_mov_b32 s45, 0\n\ts_mov_b32 s46, 0\n\ts_mov_b32 s47, 0\n\ts_mov_b32 s48, 0\n\ts_mov_b32 s49, 0\n\t" \
  "s_mov_b32 s50, 0\n\ts_mov_b32 s51, 0\n\ts_mov_b32 s52, 0\n\ts_mov_b32 s53, 0\n\ts_mov_b32 s54, 0\n\ts_mov_b32 s55, 0\n\t"

#define RATE_KERNEL(name, body)                                                          \
  __global__ void __launch_bounds__(1024) name(int iters, int spread) {                  \
    extern __shared__ char ldsb[];                                                       \
    const int lane8 = spread ? (threadIdx.x & 63) * 8 : 0;                                \
    asm volatile(PROLOGUE ::"v"(lane8) : CLOB);                                          \
    for (int i = 0; i < iters; ++i) asm volatile(".rept 8\n\t" body ".endr\n\t" ::: CLOB); \
    if (iters < 0) ldsb[threadIdx.x] = 1;                                                \
  }

// 16 entries per body (two blocks of 8, two data sets: 44.. and 24..)
RATE_KERNEL(k_old, OLD_P1("24") OLD_P2("44") "s_waitcnt lgkmcnt(0)\n\t" OLD_P1("44") OLD_P2("24") "s_waitcnt lgkmcnt(0)\n\t")
RATE_KERNEL(k_new, NEW_P1("24", 8, 9, 10, 11, 12, 13, 14, 15) NEW_P2("44", 0, 1, 2, 3, 4, 5, 6, 7, 40, 41, 42, 43) "s_waitcnt lgkmcnt(0)\n\t"
                   NEW_P1("44", 0, 1, 2, 3, 4, 5, 6, 7) NEW_P2("24", 8, 9, 10, 11, 12, 13, 14, 15, 40, 41, 42, 43) "s_waitcnt lgkmcnt(0)\n\t")
RATE_KERNEL(k_new_nolds, NEW_P2("44", 0, 1, 2, 3, 4, 5, 6, 7, 40, 41, 42, 43) NEW_P2("24", 8, 9, 10, 11, 12, 13, 14, 15, 40, 41, 42, 43))
RATE_KERNEL(k_new_p1only, NEW_P1("24", 8, 9, 10, 11, 12, 13, 14, 15) "s_waitcnt lgkmcnt(0)\n\t" NEW_P1("44", 0, 1, 2, 3, 4, 5, 6, 7) "s_waitcnt lgkmcnt(0)\n\t")
RATE_KERNEL(k_old_p1only, OLD_P1("24") "s_waitcnt lgkmcnt(0)\n\t" OLD_P1("44") "s_waitcnt lgkmcnt(0)\n\t")
RATE_KERNEL(k_old_nolds, OLD_P2("44") OLD_P2("24"))
// single-type rates, 16 instructions per body
#define R4(x) x x x x
RATE_KERNEL(k_fmac_dpp, R4("v_fmac_f32_dpp v62, v59, v24 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp v63, v59, v25 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                           "v_fmac_f32_dpp v64, v59, v26 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp v65, v59, v27 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"))
RATE_KERNEL(k_fmac_v, R4("v_fmac_f32 v62, v59, v24\n\tv_fmac_f32 v63, v59, v25\n\tv_fmac_f32 v64, v59, v26\n\tv_fmac_f32 v65, v59, v27\n\t"))
RATE_KERNEL(k_add_dpp, R4("v_add_u32_dpp v24, v58, v60 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp v25, v58, v60 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                          "v_add_u32_dpp v26, v58, v60 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp v27, v58, v60 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"))
RATE_KERNEL(k_m0, R4("s_mov_b32 m0, s40\n\ts_lshr_b32 m0, s40, 16\n\ts_mov_b32 m0, s41\n\ts_lshr_b32 m0, s41, 16\n\t"))
RATE_KERNEL(k_ds, R4("ds_read_b64 v[24:25], v60\n\tds_read_b64 v[26:27], v60\n\tds_read_b64 v[28:29], v60\n\tds_read_b64 v[30:31], v60\n\t") "s_waitcnt lgkmcnt(0)\n\t")
RATE_KERNEL(k_m0_fmac, R4("s_mov_b32 m0, s40\n\tv_fmac_f32_dpp v62, v59, v24 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp v63, v59, v25 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                          "s_lshr_b32 m0, s40, 16\n\tv_fmac_f32_dpp v64, v59, v26 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp v65, v59, v27 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"))

template <typename K>
void run(const char* name, K kern, int per_body, int threads, int spread, const char* unit) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 1000;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  kern<<<256, threads, 65536>>>(10, spread);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<<<256, threads, 65536>>>(iters, spread);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double wps = threads / 256.0;
  const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 8 * per_body * wps);
  printf("%-34s %s %2.0f waves/SIMD: %6.2f cycles per %s per SIMD (2.4 GHz)  [%s]\n", name, spread ? "lane-spread" : "same-addr  ", wps, cyc,
         unit, hipGetErrorString(hipGetLastError()));
}

int main() {
  // ---- semantics
  float* out;
  int* rows;
  hipMalloc(&out, 8 * 64 * 4);
  hipMalloc(&rows, 32);
  int h[8] = {1, 3, 1, 0, 2, 2, 3, 0};
  hipMemcpy(rows, h, 32, hipMemcpyHostToDevice);
  for (int mode : {8, 12}) {
    hipMemset(out, 0, 8 * 64 * 4);
    sem_kernel<<<1, 64>>>(out, rows, mode);
    float ho[512];
    hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost);
    // expected acc[2r + c] at lane l = sum over entries e with rows[e] == r of (2 + e) * (l + 0.25 + 0.5 c)
    int bad = 0;
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 2; ++c)
        for (int l = 0; l < 64; ++l) {
          float want = 0;
          for (int e = 0; e < 8; ++e)
            if (h[e] == r) want = fmaf(2.0f + e, l + 0.25f + 0.5f * c, want);
          if (ho[(2 * r + c) * 64 + l] != want) {
            if (bad < 4) printf("  mode %d: acc[%d] lane %d = %g, want %g\n", mode, 2 * r + c, l, ho[(2 * r + c) * 64 + l], want);
            ++bad;
          }
        }
    printf("semantics (M0[15:12] = %d): %s (%d mismatches)\n", mode, bad ? "FAIL" : "ok", bad);
  }
  {
    float* o2;
    hipMalloc(&o2, 128 * 4);
    addr_kernel<<<1, 64, 16 * 512>>>(o2);
    float ho[128];
    hipMemcpy(ho, o2, sizeof(ho), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int c = 0; c < 2; ++c)
        if (ho[l * 2 + c] != (float)(5 * 128 + l * 2 + c)) ++bad;
    printf("address path (add_dpp into the data register, ds_read_b64 from it): %s\n", bad ? "FAIL" : "ok");
  }
  // ---- rates
  for (int threads : {512, 1024}) {
    for (int spread : {0, 1}) {
      run("old mix (and_or, ds, idx, pk_fma)", k_old, 16, threads, spread, "entry");
      run("new mix (add_dpp, ds, m0, 2 fmac_dpp)", k_new, 16, threads, spread, "entry");
      run("old P1 only", k_old_p1only, 16, threads, spread, "entry");
      run("new P1 only", k_new_p1only, 16, threads, spread, "entry");
      run("ds_read_b64", k_ds, 16, threads, spread, "instr");
    }
    run("old P2 only", k_old_nolds, 16, threads, 0, "entry");
    run("new P2 only", k_new_nolds, 16, threads, 0, "entry");
    run("fmac_dpp", k_fmac_dpp, 16, threads, 0, "instr");
    run("fmac vgpr", k_fmac_v, 16, threads, 0, "instr");
    run("add_u32_dpp", k_add_dpp, 16, threads, 0, "instr");
    run("s_mov/s_lshr m0", k_m0, 16, threads, 0, "instr");
    run("m0 + 2 fmac_dpp", k_m0_fmac, 8, threads, 0, "entry");
  }
  return 0;
}
