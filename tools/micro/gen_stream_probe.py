#!/usr/bin/env python
"""Generates tools/micro/stream_probe.hip: the entry pipeline of the tiled SpMM executor in isolation (no tile DMA,
no barriers, no list heads) fed by a REAL block stream that misses the scalar cache and hits L2, in two forms:

  old: 64-byte block = 8 x (d0, value); s_load_dwordx16 per block; v_pk_fma_f32 with the value as an SGPR operand
  new: 64-byte block = 8 x d0 | 8 x value; s_load_dwordx8 of the d0 half, global_load_dwordx2 of the value half
       (lane l reads the pair at 8 * (l & 3)), two v_mfma_f32_4x4x1_16b_f32 against a register of ones broadcast the
       eight values into eight wave-uniform VGPRs (matrix pipe, no VALU slot), v_pk_fma_f32 with a VGPR multiplicand

and checks the MFMA broadcast bit for bit (denormals, -0.0, inf, NaN payloads).

    python tools/micro/gen_stream_probe.py && hipcc --offload-arch=gfx950 -O3 tools/micro/stream_probe.hip -o tools/micro/bin/stream_probe
"""
import os

RING = (40, 56, 72)          # SGPR buffers (old: 16 dwords each, new: the first 8 of each)
SET = (44, 24)               # VGPR data sets (8 register pairs each)
ADDR = (40, 41, 42, 43)
XR = (12, 14, 16, 18)        # value pairs in flight (new)
W = 4                        # v4..v11: the eight broadcast values (new)
ONES = 20


def p1(buf, dset, new):
    o = []
    for g0 in (0, 4):
        for i in range(g0, g0 + 4):
            d0 = buf + i if new else buf + 2 * i
            o.append(f"v_and_or_b32 v{ADDR[i % 4]}, s{d0}, v61, v60")
        for i in range(g0, g0 + 4):
            d = SET[dset] + 2 * i
            o.append(f"ds_read_b64 v[{d}:{d + 1}], v{ADDR[i % 4]}")
    return o


def p2(buf, dset, new):
    o = []
    for i in range(8):
        d = SET[dset] + 2 * i
        d0 = buf + i if new else buf + 2 * i
        o.append(f"s_set_gpr_idx_on s{d0}, gpr_idx(SRC2,DST)" if i == 0 else f"s_set_gpr_idx_idx s{d0}")
        if new:
            w = W + (i % 2) * 4 + i // 2          # entry i: value register (two MFMAs: even entries, odd entries)
            lo = w & ~1
            sel = 1 if w & 1 else 0
            o.append(f"v_pk_fma_f32 v[62:63], v[{d}:{d + 1}], v[{lo}:{lo + 1}], v[62:63] op_sel:[0,{sel},0] op_sel_hi:[1,{sel},1]")
        else:
            o.append(f"v_pk_fma_f32 v[62:63], v[{d}:{d + 1}], s[{buf + 2 * i}:{buf + 2 * i + 1}], v[62:63] op_sel:[0,1,0] op_sel_hi:[1,1,1]")
    o.append("s_set_gpr_idx_off")
    return o


def body(new, smem=True, fma=True, lds=True):
    """12 blocks (lcm of the 3 SGPR buffers, 2 data sets, 4 value pairs); s[36:37] = pointer of block r"""
    o = []
    for k in range(12):
        cur, nxt = RING[k % 3], RING[(k + 1) % 3]
        dc, dn = k % 2, (k + 1) % 2
        if lds:
            o += p1(nxt, dn, new)
        if fma:
            o += p2(cur, dc, new)
        if new:
            x = XR[(k + 1) % 4]
            o += ["s_waitcnt vmcnt(2)",
                  f"v_mfma_f32_4x4x1_16b_f32 v[{W}:{W + 3}], v{x}, v{ONES}, 0",
                  f"v_mfma_f32_4x4x1_16b_f32 v[{W + 4}:{W + 7}], v{x + 1}, v{ONES}, 0"]
        o += ["s_waitcnt lgkmcnt(0)"]
        if smem:
            o += [f"s_load_dwordx8 s[{cur}:{cur + 7}], s[36:37], {hex((k + 3) * 64)}" if new
                  else f"s_load_dwordx16 s[{cur}:{cur + 15}], s[36:37], {hex((k + 3) * 64)}"]
        if new:
            x = XR[k % 4]
            o += [f"global_load_dwordx2 v[{x}:{x + 1}], v21, s[36:37] offset:{(k + 4) * 64 + 32}"]
    # advance 12 blocks, wrap inside the wave's 6144-byte region (s[38:39] = region start, s34 = trips left in the region)
    o += ["s_add_u32 s36, s36, 0x300", "s_addc_u32 s37, s37, 0",
          "s_sub_u32 s34, s34, 1", "s_cmp_eq_u32 s34, 0",
          "s_cselect_b32 s36, s38, s36", "s_cselect_b32 s37, s39, s37", "s_cselect_b32 s34, 6, s34"]
    return o


def lit(lines):
    return "\n".join(f'      "{l}\\n\\t"' for l in lines)


CLOB = ", ".join(f'"v{i}"' for i in list(range(4, 22)) + list(range(24, 66)) + [127]) + ", " + \
    ", ".join(f'"s{i}"' for i in range(34, 88)) + ', "memory", "scc", "m0", "vcc"'

KERNEL = """
__global__ void __launch_bounds__(1024) {name}(int iters, const int* stream) {{
  extern __shared__ char ldsb[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  const char* region = reinterpret_cast<const char*>(stream) + (size_t)wave * REGION;
  const int lane8 = lane * 8, voff = (lane & 3) * 8;
  asm volatile(
      "v_mov_b32 v60, %1\\n\\tv_mov_b32 v61, 0xfffffe00\\n\\tv_mov_b32 v21, %2\\n\\tv_mov_b32 v{ones}, 1.0\\n\\t"
      "s_mov_b64 s[36:37], %0\\n\\ts_mov_b64 s[38:39], %0\\n\\ts_mov_b32 s34, 6\\n\\t"
      "v_mov_b32 v62, 0\\n\\tv_mov_b32 v63, 0\\n\\tv_mov_b32 v64, 0\\n\\tv_mov_b32 v65, 0\\n\\t"
{prologue}
      :: "s"(region), "v"(lane8), "v"(voff) : {clob});
  for (int i = 0; i < iters; ++i)
    asm volatile(
{body}
      ::: {clob});
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (iters < 0) ldsb[threadIdx.x] = 1;
}}
"""


def prologue(new):
    o = []
    for j, b in enumerate(RING):
        o.append(f"s_load_dwordx8 s[{b}:{b + 7}], s[36:37], {hex(j * 64)}" if new else f"s_load_dwordx16 s[{b}:{b + 15}], s[36:37], {hex(j * 64)}")
    if new:
        for j, x in enumerate(XR):
            o.append(f"global_load_dwordx2 v[{x}:{x + 1}], v21, s[36:37] offset:{j * 64 + 32}")
        o.append("s_waitcnt vmcnt(3)")
        o.append(f"v_mfma_f32_4x4x1_16b_f32 v[{W}:{W + 3}], v{XR[0]}, v{ONES}, 0")
        o.append(f"v_mfma_f32_4x4x1_16b_f32 v[{W + 4}:{W + 7}], v{XR[0] + 1}, v{ONES}, 0")
    o.append("s_waitcnt lgkmcnt(0)")
    return o


def kernel(name, new, **kw):
    return KERNEL.format(name=name, ones=ONES, prologue=lit(prologue(new)), body=lit(body(new, **kw)), clob=CLOB)


HEAD = r"""// GENERATED by tools/micro/gen_stream_probe.py - see that file.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#define REGION 6144   // bytes of block stream per wave (96 blocks), re-read in a loop: misses the scalar cache, hits L2

// MFMA broadcast: lane l holds x[l & 3] (and y[l & 3]); out[i] must be x[i] in every lane, bit for bit
__global__ void __launch_bounds__(64) bcast_kernel(const unsigned* x, unsigned* out) {
  const int lane = threadIdx.x & 63;
  unsigned v = x[lane & 3];
  unsigned r0, r1, r2, r3;
  asm volatile(
      "v_mov_b32 v20, 1.0\n\t"
      "s_nop 4\n\t"
      "v_mfma_f32_4x4x1_16b_f32 v[4:7], %4, v20, 0\n\t"
      "s_nop 15\n\t"
      "v_mov_b32 %0, v4\n\tv_mov_b32 %1, v5\n\tv_mov_b32 %2, v6\n\tv_mov_b32 %3, v7\n\t"
      : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(v) : "v4", "v5", "v6", "v7", "v20", "memory");
  out[0 * 64 + lane] = r0;
  out[1 * 64 + lane] = r1;
  out[2 * 64 + lane] = r2;
  out[3 * 64 + lane] = r3;
}
"""

MAIN = r"""
template <typename K>
void run(const char* name, K kern, int threads, const int* stream) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int iters = 400;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  kern<<<256, threads, 65536>>>(10, stream);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  kern<<<256, threads, 65536>>>(iters, stream);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double wps = threads / 256.0;
  printf("%-44s %2.0f waves/SIMD: %6.2f cycles per entry per SIMD (2.4 GHz)  [%s]\n", name, wps,
         ms * 1e-3 * 2.4e9 / ((double)iters * 96 * wps), hipGetErrorString(hipGetLastError()));
}

int main() {
  {  // ---- broadcast semantics
    const unsigned cases[][4] = {{0x3f800000u, 0x40490fdbu, 0xbf000000u, 0x00000000u},
                                 {0x80000000u, 0x00000001u, 0x807fffffu, 0x00800000u},   // -0.0, denormals, min normal
                                 {0x7f800000u, 0xff800000u, 0x7fc00000u, 0x7fa00001u},   // inf, -inf, qNaN, sNaN + payload
                                 {0x7f7fffffu, 0xff7fffffu, 0x33800000u, 0x0da24260u}};
    unsigned *dx, *dout;
    (void)hipMalloc(&dx, 16);
    (void)hipMalloc(&dout, 4 * 64 * 4);
    for (auto& c : cases) {
      (void)hipMemcpy(dx, c, 16, hipMemcpyHostToDevice);
      bcast_kernel<<<1, 64>>>(dx, dout);
      unsigned ho[256];
      (void)hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
      for (int i = 0; i < 4; ++i) {
        int bad = 0;
        for (int l = 0; l < 64; ++l) bad += ho[i * 64 + l] != c[i];
        printf("mfma broadcast of %08x: %s (lane 0 got %08x)\n", c[i], bad ? "DIFFERS" : "bit-exact", ho[i * 64]);
      }
    }
  }
  // ---- rates: 16 waves x 256 workgroups x 6144 B; d0 = 2 + 2 * row (row offsets zero), values 1.0
  const size_t waves = 256 * 16, bytes = waves * REGION + 65536;
  std::vector<int> h_old(bytes / 4, 0), h_new(bytes / 4, 0);
  for (size_t b = 0; b < bytes / 64; ++b)
    for (int i = 0; i < 8; ++i) {
      const int d0 = ((int)((b * 8 + i) % 5) << 9) | (2 + 2 * (int)((b * 3 + i) % 32));
      h_old[b * 16 + 2 * i] = d0;
      h_old[b * 16 + 2 * i + 1] = 0x3f800000;
      h_new[b * 16 + i] = d0;
      h_new[b * 16 + 8 + i] = 0x3f800000;
    }
  int *d_old, *d_new;
  (void)hipMalloc(&d_old, bytes);
  (void)hipMalloc(&d_new, bytes);
  (void)hipMemcpy(d_old, h_old.data(), bytes, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_new, h_new.data(), bytes, hipMemcpyHostToDevice);
  for (int threads : {512, 1024}) {
    run("old: x16 block, SGPR value", k_old, threads, d_old);
    run("new: x8 d0 + VMEM values + MFMA broadcast", k_new, threads, d_new);
    run("old, stream only (no LDS reads, no fma)", k_old_stream, threads, d_old);
    run("new, stream only (no LDS reads, no fma)", k_new_stream, threads, d_new);
    run("old, no scalar loads (stale SGPRs)", k_old_nosmem, threads, d_old);
    run("new, no scalar loads (stale SGPRs)", k_new_nosmem, threads, d_new);
  }
  return 0;
}
"""


def main():
    src = HEAD
    src += kernel("k_old", False) + kernel("k_new", True)
    src += kernel("k_old_stream", False, fma=False, lds=False) + kernel("k_new_stream", True, fma=False, lds=False)
    src += kernel("k_old_nosmem", False, smem=False) + kernel("k_new_nosmem", True, smem=False)
    src += MAIN
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stream_probe.hip")
    with open(p, "w") as f:
        f.write(src)


if __name__ == "__main__":
    main()
