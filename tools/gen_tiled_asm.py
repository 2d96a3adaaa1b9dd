#!/usr/bin/env python
"""Generates sparse_amd/csrc/spmm_tiled_asm.inc: the hand-scheduled inner loops of the tiled SpMM
executor (csrc/spmm_tiled.hip) as C string literals.  Run after changing the register map:

    python tools/gen_tiled_asm.py

Register map (must match spmm_tiled.hip):
    v0..v23    compiler (kernel is capped with amdgpu_num_vgpr(24))
    v24..v39   eight ds_read_b64 results, set 1 (software-pipelined loop)
    v40..v43   LDS address temporaries          v44..v59  eight ds_read_b64 results, set 0
    v60..v61   store address                    v62..v63  junk accumulator (padding entries)
    v64..v127  32 rows x (2 columns per lane) partial sums
    s36..s39   stream pointer / block counter   s40..s87  three 8-entry blocks (16 dwords each)
"""
import os

ENTRIES = 8
RING = (40, 56, 72)
ADDR = (40, 41, 42, 43)
DATA0 = 44
JUNK = 62
ROWS = 32


def round_(cur, nxt, off, pk, lds=True, fma=True, smem=True):
    o = []
    for i in range(ENTRIES if lds else 0):
        a = ADDR[i % len(ADDR)]
        o.append(f"v_and_or_b32 v{a}, s{cur + 2 * i}, %[mask], %[vbase]")
        o.append(f"ds_read_b64 v[{DATA0 + 2 * i}:{DATA0 + 2 * i + 1}], v{a}")
    o.append("s_waitcnt lgkmcnt(0)")
    # the block after next goes into the buffer that was consumed last round
    if smem:
        o.append("s_cmp_lt_u32 s38, 3")
        o.append(f"s_cbranch_scc1 7{cur}f")
        o.append(f"s_load_dwordx16 s[{nxt}:{nxt + 15}], s[36:37], {hex(off)}")
        o.append(f"7{cur}:")
    for i in range(ENTRIES if fma else 0):
        d = DATA0 + 2 * i
        if i == 0:
            o.append(f"s_set_gpr_idx_on s{cur}, gpr_idx(SRC2,DST)")
        else:
            o.append(f"s_set_gpr_idx_idx s{cur + 2 * i}")
        if pk:
            o.append(f"v_pk_fma_f32 v[{JUNK}:{JUNK + 1}], v[{d}:{d + 1}], s[{cur + 2 * i}:{cur + 2 * i + 1}], "
                     f"v[{JUNK}:{JUNK + 1}] op_sel:[0,1,0] op_sel_hi:[1,1,1]")
        else:
            o.append(f"v_fma_f32 v{JUNK}, s{cur + 2 * i + 1}, v{d}, v{JUNK}")
            o.append(f"v_fma_f32 v{JUNK + 1}, s{cur + 2 * i + 1}, v{d + 1}, v{JUNK + 1}")
    if fma:
        o.append("s_set_gpr_idx_off")
    return o


def consume(pk, lds=True, fma=True, smem=True):
    A, B, C = RING
    o = ["s_mov_b64 s[36:37], %[ptr]",
         "s_mov_b32 s38, %[nblk]",
         f"s_load_dwordx16 s[{A}:{A + 15}], s[36:37], 0x0",
         "s_cmp_lt_u32 s38, 2",
         "s_cbranch_scc1 10f" if smem else "s_nop 0",
         f"s_load_dwordx16 s[{B}:{B + 15}], s[36:37], 0x40",
         "10:",
         "s_waitcnt lgkmcnt(0)",
         "11:"]
    if not smem:  # ablation: blocks are fetched once per list and reused (wrong results, same instruction mix)
        o.insert(-2, f"s_load_dwordx16 s[{C}:{C + 15}], s[36:37], 0x0")
    for cur, nxt, off in ((A, C, 0x80), (B, A, 0xC0), (C, B, 0x100)):
        o += round_(cur, nxt, off, pk, lds, fma, smem)
        o += ["s_sub_u32 s38, s38, 1", "s_cmp_eq_u32 s38, 0", "s_cbranch_scc1 12f"]
    o += ["s_add_u32 s36, s36, 0xc0", "s_addc_u32 s37, s37, 0", "s_branch 11b", "12:"]
    return o


DATASET = (44, 24)


def p1(buf, dset):
    """addresses + LDS reads of the block held in SGPR buffer `buf` into VGPR data set `dset`"""
    o = []
    for i in range(ENTRIES):
        a = ADDR[i % len(ADDR)]
        d = DATASET[dset] + 2 * i
        o.append(f"v_and_or_b32 v{a}, s{buf + 2 * i}, %[mask], %[vbase]")
        o.append(f"ds_read_b64 v[{d}:{d + 1}], v{a}")
    return o


def p2(buf, dset):
    """the eight fused multiply-adds of a block whose B rows are in data set `dset`"""
    o = []
    for i in range(ENTRIES):
        d = DATASET[dset] + 2 * i
        o.append(f"s_set_gpr_idx_on s{buf}, gpr_idx(SRC2,DST)" if i == 0 else f"s_set_gpr_idx_idx s{buf + 2 * i}")
        o.append(f"v_pk_fma_f32 v[{JUNK}:{JUNK + 1}], v[{d}:{d + 1}], s[{buf + 2 * i}:{buf + 2 * i + 1}], "
                 f"v[{JUNK}:{JUNK + 1}] op_sel:[0,1,0] op_sel_hi:[1,1,1]")
    o.append("s_set_gpr_idx_off")
    return o


def consume_pipelined():
    """Software-pipelined list loop: while block r is multiplied (P2), the B rows of block r+1 are already
    being read from LDS (P1) and block r+3 is being fetched by a scalar load.  s38 = blocks left
    (including the one whose P2 is next); unrolled x6 = lcm(3 SGPR buffers, 2 VGPR data sets)."""
    A, B, C = RING
    o = ["s_mov_b64 s[36:37], %[ptr]",
         "s_mov_b32 s38, %[nblk]",
         f"s_load_dwordx16 s[{A}:{A + 15}], s[36:37], 0x0",
         "s_cmp_lt_u32 s38, 2",
         "s_cbranch_scc1 10f",
         f"s_load_dwordx16 s[{B}:{B + 15}], s[36:37], 0x40",
         "s_cmp_lt_u32 s38, 3",
         "s_cbranch_scc1 10f",
         f"s_load_dwordx16 s[{C}:{C + 15}], s[36:37], 0x80",
         "10:",
         "s_waitcnt lgkmcnt(0)"]
    o += p1(A, 0)
    o += ["s_waitcnt lgkmcnt(0)", "11:"]
    for k in range(6):
        cur, nxt = RING[k % 3], RING[(k + 1) % 3]
        dc, dn = k % 2, (k + 1) % 2
        off = (k + 3) * 64
        # fast path: at least 4 blocks left -> block r+1 exists and block r+3 is fetched
        o += ["s_cmp_lt_u32 s38, 4", f"s_cbranch_scc1 3{k}f"]
        o += p1(nxt, dn) + p2(cur, dc)
        o += ["s_waitcnt lgkmcnt(0)", f"s_load_dwordx16 s[{cur}:{cur + 15}], s[36:37], {hex(off)}",
              "s_sub_u32 s38, s38, 1", f"s_branch 4{k}f"]
        # slow path: the last three blocks of the list
        o += [f"3{k}:", "s_cmp_lt_u32 s38, 2", f"s_cbranch_scc1 5{k}f"]
        o += p1(nxt, dn)
        o += [f"5{k}:"] + p2(cur, dc)
        o += ["s_waitcnt lgkmcnt(0)", "s_sub_u32 s38, s38, 1", "s_cmp_eq_u32 s38, 0", "s_cbranch_scc1 12f", f"4{k}:"]
    o += ["s_add_u32 s36, s36, 0x180", "s_addc_u32 s37, s37, 0", "s_branch 11b", "12:"]
    return o


def store():
    o = ["v_mov_b32 v60, %[lo]", "v_mov_b32 v61, %[hi]", "s_mov_b32 s36, 0"]
    for j in range(ROWS):
        o += ["s_cmp_ge_i32 s36, %[n]", "s_cbranch_scc1 9f",
              f"global_store_dwordx2 v[60:61], v[{64 + 2 * j}:{65 + 2 * j}], off nt",
              "v_lshl_add_u64 v[60:61], %[stride], 0, v[60:61]",
              "s_add_i32 s36, s36, 1"]
    o.append("9:")
    return o


def zero():
    return [f"v_mov_b32 v{r}, 0" for r in range(JUNK, 128)]


def lit(name, lines):
    body = "\n".join(f'  "{l}\\n\\t"' for l in lines)
    return f"#define {name} \\\n" + body.replace("\n", " \\\n") + "\n"


def clob(prefix, lo, hi):
    return ", ".join(f'"{prefix}{i}"' for i in range(lo, hi + 1))


def main():
    out = ["// GENERATED by tools/gen_tiled_asm.py - do not edit.\n",
           lit("TL_ASM_CONSUME", consume(False)),
           lit("TL_ASM_CONSUME_PK", consume(True)),
           lit("TL_ASM_CONSUME_PIPE", consume_pipelined()),
           lit("TL_ASM_CONSUME_NOFMA", consume(False, True, False)),
           lit("TL_ASM_CONSUME_NOLDS", consume(False, False, False)),
           lit("TL_ASM_CONSUME_NOSMEM", consume(True, True, True, False)),
           lit("TL_ASM_STORE", store()),
           lit("TL_ASM_ZERO", zero()),
           f"#define TL_CLOB_SGPR {clob('s', 36, 87)}\n",
           f"#define TL_CLOB_TMP {clob('v', 24, 61)}\n",
           f"#define TL_CLOB_ACC {clob('v', 62, 127)}\n"]
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sparse_amd", "csrc", "spmm_tiled_asm.inc")
    with open(p, "w") as f:
        f.write("\n".join(out))


if __name__ == "__main__":
    main()
