#!/usr/bin/env python
"""Generates sparse_amd/csrc/spmm_tiled_asm.inc: the hand-scheduled inner loops of the tiled SpMM
executor (csrc/spmm_tiled.hip) as C string literals.  Run after changing the register map:

    python tools/gen_tiled_asm.py

Register map (must match spmm_tiled.hip):
    v0..v21    compiler (kernel is capped with amdgpu_num_vgpr(22))
    v22..v23   walking source pointer of the wave's tile-DMA share
    v24..v39   eight ds_read_b64 results, set 1 (software-pipelined loop)
    v40..v43   LDS address temporaries          v44..v59  eight ds_read_b64 results, set 0
    v60        LDS base of the current tile + 8*lane    v61  destination of the line touches
    (v60..v61  store address after the loop)     v62..v63  junk accumulator (padding entries)
    v64..v127  32 rows x (2 columns per lane) partial sums
    s36..s39   stream pointer / block and DMA counters   s40..s87  three 8-entry blocks (16 dwords each)
    s88..s89   temporary / LDS destination of the next tile-DMA instruction
    s90..s95   tile counter, first blocks of lists t, t+1, t+2, 64-bit temporary
"""
import os

ENTRIES = 8      # entries per 16-dword block: 8 x (d0, f32 value) or, for f64, 5 d0 + pad + 5 x (value lo, hi)
F64 = False
RING = (40, 56, 72)
ADDR = (40, 41, 42, 43)
DATA0 = 44
JUNK = 62
ROWS = 32


DATASET = (44, 24)


def d0_reg(buf, i):
    return buf + i if F64 else buf + 2 * i


def val_pair(buf, i):
    """SGPR pair holding the value of entry i (f64: the value itself; f32: (d0, value), selected with op_sel)"""
    return (buf + 6 + 2 * i, buf + 7 + 2 * i) if F64 else (buf + 2 * i, buf + 2 * i + 1)


def p1(buf, dset):
    """addresses + LDS reads of the block held in SGPR buffer `buf` into VGPR data set `dset`"""
    o = []
    for i in range(ENTRIES):
        a = ADDR[i % len(ADDR)]
        d = DATASET[dset] + 2 * i
        o.append(f"v_and_or_b32 v{a}, s{d0_reg(buf, i)}, %[mask], v60")
        o.append(f"ds_read_b64 v[{d}:{d + 1}], v{a}")
    return o


def p2(buf, dset):
    """the eight fused multiply-adds of a block whose B rows are in data set `dset`"""
    o = []
    for i in range(ENTRIES):
        d = DATASET[dset] + 2 * i
        lo, hi = val_pair(buf, i)
        o.append(f"s_set_gpr_idx_on s{d0_reg(buf, 0)}, gpr_idx(SRC2,DST)" if i == 0
                 else f"s_set_gpr_idx_idx s{d0_reg(buf, i)}")
        if F64:
            o.append(f"v_fma_f64 v[{JUNK}:{JUNK + 1}], s[{lo}:{hi}], v[{d}:{d + 1}], v[{JUNK}:{JUNK + 1}]")
        else:
            o.append(f"v_pk_fma_f32 v[{JUNK}:{JUNK + 1}], v[{d}:{d + 1}], s[{lo}:{hi}], "
                     f"v[{JUNK}:{JUNK + 1}] op_sel:[0,1,0] op_sel_hi:[1,1,1]")
    o.append("s_set_gpr_idx_off")
    return o


def p2_exact(buf, dset):
    """reference arithmetic (`out[i, j] += data * b[k, j]`, _common.py:752): a rounded product, then a rounded
    add - the products overwrite the B rows in place, only the adds run under the gpr-index mode"""
    o = []
    for i in range(ENTRIES):
        d = DATASET[dset] + 2 * i
        lo, hi = val_pair(buf, i)
        if F64:
            o.append(f"v_mul_f64 v[{d}:{d + 1}], v[{d}:{d + 1}], s[{lo}:{hi}]")
        else:
            o.append(f"v_pk_mul_f32 v[{d}:{d + 1}], v[{d}:{d + 1}], s[{lo}:{hi}] op_sel:[0,1] op_sel_hi:[1,1]")
    for i in range(ENTRIES):
        d = DATASET[dset] + 2 * i
        o.append(f"s_set_gpr_idx_on s{d0_reg(buf, 0)}, gpr_idx(SRC1,DST)" if i == 0
                 else f"s_set_gpr_idx_idx s{d0_reg(buf, i)}")
        if F64:
            o.append(f"v_add_f64 v[{JUNK}:{JUNK + 1}], v[{d}:{d + 1}], v[{JUNK}:{JUNK + 1}]")
        else:
            o.append(f"v_pk_add_f32 v[{JUNK}:{JUNK + 1}], v[{d}:{d + 1}], v[{JUNK}:{JUNK + 1}]")
    o.append("s_set_gpr_idx_off")
    return o


def dma_hook(site):
    """Issue one more tile-DMA instruction of the NEXT tile if any are left (s39), from the walking source
    pointer v[22:23] (advanced by 32 B rows) to LDS address s89 (advanced by 16 KB)."""
    return ["s_cmp_eq_u32 s39, 0",
            f"s_cbranch_scc1 {60 + site}f",
            "s_mov_b32 m0, s89",
            "s_nop 0",
            "global_load_lds_dwordx4 v[22:23], off",
            "s_add_u32 s89, s89, 0x4000",
            "v_lshl_add_u64 v[22:23], %[step], 0, v[22:23]",
            "s_sub_u32 s39, s39, 1",
            f"{60 + site}:"]


def list_loop(lds=True, fma=True, exact=False):
    """The list loop of one tile phase.  Software-pipelined: while block r is multiplied (P2), the B rows
    of block r+1 are already being read from LDS (P1) and block r+3 is being fetched by a scalar load.
    Entry: s[36:37] = list pointer, s38 = blocks in the list (> 0), the first min(3, s38) blocks are in
    (or on their way into) the SGPR ring.  Unrolled x6 = lcm(3 SGPR buffers, 2 VGPR data sets).  The wave's
    share of the NEXT tile's LDS-DMA (s39 instructions) is spread over the rounds (one at the start, one
    after each round; the caller issues the rest) instead of one burst that would stall all 16 waves on
    the 64 B/clk address path."""
    A, B, C = RING
    P1 = p1 if lds else (lambda buf, dset: [])
    P2 = (p2_exact if exact else p2) if fma else (lambda buf, dset: [])
    o = dma_hook(0)
    o += ["s_waitcnt lgkmcnt(0)"]
    o += P1(A, 0)
    o += ["s_waitcnt lgkmcnt(0)", "11:"]
    for k in range(6):
        cur, nxt = RING[k % 3], RING[(k + 1) % 3]
        dc, dn = k % 2, (k + 1) % 2
        off = (k + 3) * 64
        # fast path: at least 4 blocks left -> block r+1 exists and block r+3 is fetched
        o += ["s_cmp_lt_u32 s38, 4", f"s_cbranch_scc1 3{k}f"]
        o += P1(nxt, dn) + P2(cur, dc)
        o += ["s_waitcnt lgkmcnt(0)", f"s_load_dwordx16 s[{cur}:{cur + 15}], s[36:37], {hex(off)}"]
        o += dma_hook(1 + 2 * k)
        o += ["s_sub_u32 s38, s38, 1", f"s_branch 4{k}f"]
        # slow path: the last three blocks of the list
        o += [f"3{k}:", "s_cmp_lt_u32 s38, 2", f"s_cbranch_scc1 5{k}f"]
        o += P1(nxt, dn)
        o += [f"5{k}:"] + P2(cur, dc)
        o += ["s_waitcnt lgkmcnt(0)"]
        o += dma_hook(2 + 2 * k)
        o += ["s_sub_u32 s38, s38, 1", "s_cmp_eq_u32 s38, 0", "s_cbranch_scc1 12f", f"4{k}:"]
    o += ["s_add_u32 s36, s36, 0x180", "s_addc_u32 s37, s37, 0", "s_branch 11b"]
    return o


def list_pointer(first_block_sgpr):
    """s[36:37] = stream base + 64 * first block"""
    return [f"s_mov_b32 s94, s{first_block_sgpr}", "s_mov_b32 s95, 0", "s_lshl_b64 s[94:95], s[94:95], 6",
            "s_add_u32 s36, s94, %[blo]", "s_addc_u32 s37, s95, %[bhi]"]


def request_first_blocks(lo, hi, tag):
    """scalar loads of the first min(3, s{hi} - s{lo}) blocks of a list into ring buffers A, B, C (no wait)"""
    A, B, C = RING
    return list_pointer(lo) + [
        f"s_sub_u32 s88, s{hi}, s{lo}",
        "s_cmp_eq_u32 s88, 0", f"s_cbranch_scc1 {tag}f",
        f"s_load_dwordx16 s[{A}:{A + 15}], s[36:37], 0x0",
        "s_cmp_lt_u32 s88, 2", f"s_cbranch_scc1 {tag}f",
        f"s_load_dwordx16 s[{B}:{B + 15}], s[36:37], 0x40",
        "s_cmp_lt_u32 s88, 3", f"s_cbranch_scc1 {tag}f",
        f"s_load_dwordx16 s[{C}:{C + 15}], s[36:37], 0x80",
        f"{tag}:"]


def phases(lds=True, fma=True, exact=False):
    """Tile phases t0 .. te-1 of one wave in ONE asm block, so that SGPR state survives the barrier:
    the first blocks of list t+1 are requested BEFORE the barrier that ends phase t (the scalar path serves
    one 64-byte request per ~20 cycles per CU; 16 waves x 3 requests right after a barrier idle the CU for
    ~1000 cycles).  s90 = t, s91/s92/s93 = first block of lists t, t+1, t+2."""
    o = ["s_mov_b32 s90, %[t0]", "s_mov_b32 s91, %[o0]", "s_mov_b32 s92, %[o1]", "s_mov_b32 s93, %[o2]"]
    o += request_first_blocks(91, 92, 20)
    o += ["1:"]
    o += list_pointer(91)
    o += ["s_sub_u32 s38, s92, s91",                       # blocks in this list
          "s_add_u32 s88, s90, 1",                         # next tile: DMA share and LDS destination
          "s_cmp_lt_u32 s88, %[nfull]", "s_cselect_b32 s39, 4, 0",
          "s_and_b32 s88, s88, 1", "s_lshl_b32 s88, s88, 16", "s_add_u32 s89, s88, %[m0wave]",
          "s_and_b32 s88, s90, 1", "s_lshl_b32 s88, s88, 16", "v_or_b32 v60, s88, %[lane8]",  # this tile's LDS base
          "s_cmp_eq_u32 s38, 0", "s_cbranch_scc1 12f"]
    o += list_loop(lds, fma, exact)
    o += ["12:", "13:"] + dma_hook(13)[:-1] + ["s_branch 13b", "73:"]   # the rest of the DMA share
    # first blocks of the next list (not at the end of the chunk: the ring must be idle when the asm ends)
    o += ["s_add_u32 s88, s90, 1", "s_cmp_lt_u32 s88, %[te]", "s_cbranch_scc0 21f"]
    o += request_first_blocks(92, 93, 21)
    # o3 = first block of list t+3 (lane min(t+3, ntiles) - obase of the offsets register)
    o += ["s_add_u32 s88, s90, 3", "s_min_u32 s88, s88, %[ntiles]", "s_sub_u32 s88, s88, %[obase]", "s_nop 3",
          "v_readlane_b32 s88, %[offreg], s88", "s_nop 3"]
    # touch the lines of list t+2 = blocks [s93, s88): lane i -> line min(i, n-1)  (always ONE instruction)
    o += ["s_sub_i32 s94, s88, s93", "s_sub_i32 s94, s94, 1", "s_max_i32 s94, s94, 0",
          "v_min_u32 v42, s94, %[lane]", "v_lshlrev_b32 v42, 6, v42", "v_mov_b32 v43, 0",
          "s_mov_b32 s94, s93", "s_mov_b32 s95, 0", "s_lshl_b64 s[94:95], s[94:95], 6",
          "s_add_u32 s94, s94, %[blo]", "s_addc_u32 s95, s95, %[bhi]",
          "v_lshl_add_u64 v[40:41], s[94:95], 0, v[42:43]",
          "global_load_dword v61, v[40:41], off"]
    o += ["s_mov_b32 s91, s92", "s_mov_b32 s92, s93", "s_mov_b32 s93, s88",
          "s_waitcnt vmcnt(1)",      # tile t+1 (this wave's share) has landed; the touch may still fly
          "s_barrier",
          "s_add_u32 s90, s90, 1", "s_cmp_lt_u32 s90, %[te]", "s_cbranch_scc1 1b",
          "s_waitcnt lgkmcnt(0)"]
    return o


def tile0():
    """the wave's four DMA instructions of tile 0 (buffer 0), advancing the walking pointer"""
    o = ["s_mov_b32 s89, %[m0wave]", "s_mov_b32 s39, 4"]
    return o + ["13:"] + dma_hook(13)[:-1] + ["s_branch 13b", "73:"]


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


def f64_variants():
    """the same phases for float64: one column per lane (v_fma_f64 on a register pair), 5 entries per block"""
    global ENTRIES, F64
    ENTRIES, F64 = 5, True
    try:
        return [lit("TL_ASM_PHASES_F64", phases()), lit("TL_ASM_PHASES_F64_EXACT", phases(True, True, True))]
    finally:
        ENTRIES, F64 = 8, False


def main():
    out = ["// GENERATED by tools/gen_tiled_asm.py - do not edit.\n",
           lit("TL_ASM_PHASES", phases()),
           lit("TL_ASM_PHASES_EXACT", phases(True, True, True)),
           lit("TL_ASM_PHASES_NOFMA", phases(True, False)),
           lit("TL_ASM_PHASES_NOLDS", phases(False, False)),
           *f64_variants(),
           lit("TL_ASM_TILE0", tile0()),
           lit("TL_ASM_STORE", store()),
           lit("TL_ASM_ZERO", zero()),
           f"#define TL_CLOB_SGPR {clob('s', 36, 95)}\n",
           f"#define TL_CLOB_TMP {clob('v', 22, 61)}\n",
           f"#define TL_CLOB_ACC {clob('v', 62, 127)}\n"]
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sparse_amd", "csrc", "spmm_tiled_asm.inc")
    with open(p, "w") as f:
        f.write("\n".join(out))


if __name__ == "__main__":
    main()
