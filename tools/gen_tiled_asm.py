#!/usr/bin/env python
"""Generates sparse_amd/csrc/spmm_tiled_asm.inc: the hand-scheduled inner loops of the tiled SpMM
executor (csrc/spmm_tiled.hip) as C string literals.  Run after changing the register map:

    python tools/gen_tiled_asm.py

Register map (must match spmm_tiled.hip):
    v0..v21    compiler (kernel is capped with amdgpu_num_vgpr(22))
    (the walking source pointer of the wave's tile-DMA share is an in/out operand: the compiler keeps it)
    v22        LDS base of the current tile + 8*lane    v23  destination of the line touches
    (v22..v23  store address after the loop)
    v24..v39   eight ds_read_b64 results, set 1 (software-pipelined loop)
    v40..v55   eight ds_read_b64 results, set 0
               (an entry's LDS address is computed INTO the low register of its result pair: no address temporaries,
               which is what makes room for 35 instead of 32 rows per wave)
    v56..v57   junk accumulator (padding entries)
    v58..v127  35 rows x (2 columns per lane) partial sums
    s36..s38   stream pointer / block counter (s39 free; pending DMA instructions: VCC as a mask of ones)
    s40..s87   three 8-entry blocks (16 dwords each)
    s88..s89   temporary / LDS destination of the next tile-DMA instruction
    s90..s95   tile counter, first blocks of lists t, t+1, t+2, 64-bit temporary
"""
import os

ENTRIES = 8      # entries per 16-dword block: 8 x (d0, f32 value) or, for f64, 5 d0 + pad + 5 x (value lo, hi)
F64 = False
I32 = False      # int32 values and B (wrap-around arithmetic): the float32 layout, the "exact" structure with integer opcodes
RING = (40, 56, 72)
ROWS = int(os.environ.get("TL_RG", "35"))     # rows per wave: 2 accumulator registers each, v[128 - 2*ROWS : 128)
WAVES = int(os.environ.get("TL_WAVES", "16"))  # waves per workgroup (ROWS * WAVES rows share one B tile)
VB = int(os.environ.get("TL_VGPRS", "128"))    # registers of a wave (128: four waves per SIMD; 168: three)
ACC0 = VB - 2 * ROWS
JUNK = ACC0 - 2                               # junk accumulator pair (padding entries), just below the accumulators
COMP = int(os.environ.get("TL_COMP", "22"))   # v0 .. v(COMP-1) belong to the compiler (amdgpu_num_vgpr)
BASE = COMP                                   # LDS base of the current tile + 8*lane
TOUCH = COMP + 1                              # destination of the line touches


DATASET = (COMP + 18, COMP + 2)
assert JUNK >= DATASET[0] + 16, "accumulators collide with the data sets"


def d0_reg(buf, i):
    return buf + i if F64 else buf + 2 * i


def val_pair(buf, i):
    """SGPR pair holding the value of entry i (f64: the value itself; f32: (d0, value), selected with op_sel)"""
    return (buf + 6 + 2 * i, buf + 7 + 2 * i) if F64 else (buf + 2 * i, buf + 2 * i + 1)


P1_GROUP = int(os.environ.get("TL_P1_GROUP", "4"))          # address computations issued back to back before their LDS reads
DMA_AT_START = int(os.environ.get("TL_DMA_AT_START", "2"))  # tile-DMA instructions issued before the first block of a list
KB = int(os.environ.get("TL_KB", "160"))                    # B rows per LDS tile (multiple of 32; 2 x KB x 512 B <= 160 KB)
DMA_PER_TILE = KB // (2 * WAVES)                             # LDS-DMA instructions per wave per tile
PENDING = (1 << DMA_PER_TILE) - 1                            # VCC mask of a full tile's pending DMA instructions
STAGE = int(os.environ.get("TL_STAGE", "0"))   # 1: B tiles by global loads into four staging registers + ds_write_b128 instead of LDS-DMA
STG = 18                                        # v18..v21 staging (an even-aligned tuple), v17 LDS write address (clobbers: the compiler keeps nothing there across the asm)
WADDR = 17
HOOK_BEFORE_WAIT = int(os.environ.get("TL_HOOK_BEFORE_WAIT", "0"))  # the list loop's DMA hook ahead of the block's wait (the issue stall overlaps the wait)
TAIL_HOOKS = int(os.environ.get("TL_TAIL_HOOKS", "0"))      # DMA hooks inside the three-block tails (the rest waits for the list end)


def p1(buf, dset):
    """addresses + LDS reads of the block held in SGPR buffer `buf` into VGPR data set `dset`"""
    o = []
    for g0 in range(0, ENTRIES, P1_GROUP):
        grp = range(g0, min(g0 + P1_GROUP, ENTRIES))
        for i in grp:
            d = DATASET[dset] + 2 * i
            o.append(f"v_and_or_b32 v{d}, s{d0_reg(buf, i)}, %[mask], v{BASE}")
        for i in grp:
            d = DATASET[dset] + 2 * i
            o.append(f"ds_read_b64 v[{d}:{d + 1}], v{d}")
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


HALF_SKIP = int(os.environ.get("TL_HALF_SKIP", "1"))   # the last block of a list skips its upper half when that is all padding


def _with_half_skip(lines_lo, lines_hi, buf, label):
    """`lines_lo` (entries below ENTRIES // 2) always, `lines_hi` only if the block's entry ENTRIES // 2 is a real one: padding
    entries have d0 = 0 (a real d0 carries an accumulator index >= 2), and a list's entries are packed from the front, so
    d0[ENTRIES // 2] == 0 means the whole upper half is padding."""
    if not HALF_SKIP:
        return lines_lo + lines_hi
    return lines_lo + [f"s_cmp_eq_u32 s{d0_reg(buf, ENTRIES // 2)}, 0", f"s_cbranch_scc1 {label}f"] + lines_hi + [f"{label}:"]


def p1_last(buf, dset, label):
    """p1 for the LAST block of a list"""
    half = ENTRIES // 2
    lo, hi = [], []
    for part, rng in ((lo, range(0, half)), (hi, range(half, ENTRIES))):
        for i in rng:
            d = DATASET[dset] + 2 * i
            part.append(f"v_and_or_b32 v{d}, s{d0_reg(buf, i)}, %[mask], v{BASE}")
        for i in rng:
            d = DATASET[dset] + 2 * i
            part.append(f"ds_read_b64 v[{d}:{d + 1}], v{d}")
    return _with_half_skip(lo, hi, buf, label)


def p2_last(buf, dset, label, exact=False):
    """p2 / p2_exact for the LAST block of a list.  Exact mode: the products (no index mode) and the adds (under the index
    mode) are two separate runs over the entries, each with its own skip of the upper half."""
    half = ENTRIES // 2
    lo_r, hi_r = list(range(0, half)), list(range(half, ENTRIES))

    def muls(rng):
        o = []
        for i in rng:
            d = DATASET[dset] + 2 * i
            lo_, hi_ = val_pair(buf, i)
            if I32:
                o += int_muls(d, hi_)
                continue
            o.append(f"v_mul_f64 v[{d}:{d + 1}], v[{d}:{d + 1}], s[{lo_}:{hi_}]" if F64 else
                     f"v_pk_mul_f32 v[{d}:{d + 1}], v[{d}:{d + 1}], s[{lo_}:{hi_}] op_sel:[0,1] op_sel_hi:[1,1]")
        return o

    def accs(rng, first):
        o = []
        for i in rng:
            d = DATASET[dset] + 2 * i
            lo_, hi_ = val_pair(buf, i)
            mode = "gpr_idx(SRC1,DST)" if exact else "gpr_idx(SRC2,DST)"
            o.append(f"s_set_gpr_idx_on s{d0_reg(buf, i)}, {mode}" if (first and i == rng[0]) else f"s_set_gpr_idx_idx s{d0_reg(buf, i)}")
            if exact and I32:
                o += int_adds(d)
            elif exact:
                o.append(f"v_add_f64 v[{JUNK}:{JUNK + 1}], v[{d}:{d + 1}], v[{JUNK}:{JUNK + 1}]" if F64 else
                         f"v_pk_add_f32 v[{JUNK}:{JUNK + 1}], v[{d}:{d + 1}], v[{JUNK}:{JUNK + 1}]")
            elif F64:
                o.append(f"v_fma_f64 v[{JUNK}:{JUNK + 1}], s[{lo_}:{hi_}], v[{d}:{d + 1}], v[{JUNK}:{JUNK + 1}]")
            else:
                o.append(f"v_pk_fma_f32 v[{JUNK}:{JUNK + 1}], v[{d}:{d + 1}], s[{lo_}:{hi_}], "
                         f"v[{JUNK}:{JUNK + 1}] op_sel:[0,1,0] op_sel_hi:[1,1,1]")
        return o

    o = _with_half_skip(muls(lo_r), muls(hi_r), buf, label) if exact else []
    return o + _with_half_skip(accs(lo_r, True), accs(hi_r, False), buf, label) + ["s_set_gpr_idx_off"]


def p2_exact(buf, dset):
    """reference arithmetic (`out[i, j] += data * b[k, j]`, _common.py:752): a rounded product, then a rounded
    add - the products overwrite the B rows in place, only the adds run under the gpr-index mode"""
    o = []
    for i in range(ENTRIES):
        d = DATASET[dset] + 2 * i
        lo, hi = val_pair(buf, i)
        if F64:
            o.append(f"v_mul_f64 v[{d}:{d + 1}], v[{d}:{d + 1}], s[{lo}:{hi}]")
        elif I32:
            o += int_muls(d, hi)
        else:
            o.append(f"v_pk_mul_f32 v[{d}:{d + 1}], v[{d}:{d + 1}], s[{lo}:{hi}] op_sel:[0,1] op_sel_hi:[1,1]")
    for i in range(ENTRIES):
        d = DATASET[dset] + 2 * i
        o.append(f"s_set_gpr_idx_on s{d0_reg(buf, 0)}, gpr_idx(SRC1,DST)" if i == 0
                 else f"s_set_gpr_idx_idx s{d0_reg(buf, i)}")
        if F64:
            o.append(f"v_add_f64 v[{JUNK}:{JUNK + 1}], v[{d}:{d + 1}], v[{JUNK}:{JUNK + 1}]")
        elif I32:
            o += int_adds(d)
        else:
            o.append(f"v_pk_add_f32 v[{JUNK}:{JUNK + 1}], v[{d}:{d + 1}], v[{JUNK}:{JUNK + 1}]")
    o.append("s_set_gpr_idx_off")
    return o


def int_muls(d, sval):
    """two columns per lane: the low 32 bits of column x value, in place (wrap-around: what NumPy's int32 product is)"""
    return [f"v_mul_lo_u32 v{d}, v{d}, s{sval}", f"v_mul_lo_u32 v{d + 1}, v{d + 1}, s{sval}"]


def int_adds(d):
    """under gpr_idx(SRC1,DST): accumulator pair += products (the index offsets vsrc1 and vdst, not src0)"""
    return [f"v_add_u32 v{JUNK}, v{d}, v{JUNK}", f"v_add_u32 v{JUNK + 1}, v{d + 1}, v{JUNK + 1}"]


def dma_body():
    """one tile-DMA instruction of the NEXT tile, in the buffer form: source = the B descriptor %[srd] (SGPRs) + the lane's
    fixed byte offset %[voff] + the walking scalar offset %[soff] (a compiler-owned in/out SGPR, advanced one round of B
    rows by a SALU add: the flat form needed a 64-bit VALU add on a per-lane pointer pair and twice the address registers);
    LDS address s89 (advanced by 16 row pairs = 32 KB of the interleaved buffers; the s_add also is the wait state
    M0 needs before an LDS-DMA), and one bit less in the pending mask (VCC)"""
    prio = int(os.environ.get("TL_DMA_PRIO", "0"))
    return (["s_setprio %d" % prio] if prio else []) + [
            "s_mov_b32 m0, s89",
            f"s_add_u32 s89, s89, {hex(WAVES * 2048)}",
            "buffer_load_dwordx4 %[voff], %[srd], %[soff] offen lds",
            "s_add_u32 %[soff], %[soff], %[step]",
            "s_lshr_b64 vcc, vcc, 1"] + (["s_setprio 0"] if prio else [])


def stage_flush():
    """the staged 1 KB piece (v17..v20, loaded by an earlier hook) goes to LDS at s89 + 16 * lane"""
    return ["s_waitcnt vmcnt(0)",
            f"v_lshl_add_u32 v{WADDR}, %[lane8], 1, s89",
            f"ds_write_b128 v{WADDR}, v[{STG}:{STG + 3}]",
            f"s_add_u32 s89, s89, {hex(WAVES * 2048)}",
            "s_mov_b32 s39, 0"]


def stage_issue(q=STG):
    return [f"buffer_load_dwordx4 v[{q}:{q + 3}], %[voff], %[srd], %[soff] offen",
            "s_add_u32 %[soff], %[soff], %[step]"]


def dma_hook():
    """Issue one more tile-DMA instruction if any are left.  The pending count lives in VCC as a mask of ones so
    that the test is a single s_cbranch_vccz (SCC is clobbered when an instruction is issued).
    Staged form: the piece loaded by the PREVIOUS hook (a block ago: an L2 round trip has passed) is written to LDS,
    then the next piece is requested; s39 = a piece is staged."""
    if not STAGE:
        return ["s_cbranch_vccz 6f"] + dma_body() + ["6:"]
    if STAGE == 2:   # every piece of the next tile is requested, waited for and written at the END of the list (dma_rest)
        return []
    return (["s_cmp_eq_u32 s39, 0", "s_cbranch_scc1 5f"] + stage_flush() + ["5:", "s_cbranch_vccz 6f"] + stage_issue() +
            ["s_lshr_b64 vcc, vcc, 1", "s_mov_b32 s39, 1", "6:"])


def dma_rest():
    """what is left of the wave's share at the end of its list.  Staged form: the staged piece is flushed, then ALL
    remaining pieces are requested into the (now idle) data-set registers, waited for once and written."""
    if not STAGE:
        return ["13:", "s_cbranch_vccz 14f"] + dma_body() + ["s_branch 13b", "14:"]
    o = ["s_cmp_eq_u32 s39, 0", "s_cbranch_scc1 5f"] + stage_flush() + ["5:", "s_cbranch_vccz 14f", "s_bcnt1_i32_b64 s38, vcc"]
    n = KB // (2 * WAVES)
    for i in range(n):
        o += [f"s_cmp_le_u32 s38, {i}", "s_cbranch_scc1 13f"] + stage_issue(DATASET[1] + 4 * i)
    o += ["13:", "s_waitcnt vmcnt(0)"]
    for i in range(n):
        q = DATASET[1] + 4 * i
        o += [f"s_cmp_le_u32 s38, {i}", "s_cbranch_scc1 14f",
              f"v_lshl_add_u32 v{WADDR}, %[lane8], 1, s89", f"ds_write_b128 v{WADDR}, v[{q}:{q + 3}]",
              f"s_add_u32 s89, s89, {hex(WAVES * 2048)}"]
    o += ["14:", "s_mov_b64 vcc, 0"]
    return o


def list_loop(lds=True, fma=True, exact=False):
    """The list loop of one tile phase.  Software-pipelined: while block r is multiplied (P2), the B rows
    of block r+1 are already being read from LDS (P1) and block r+3 is being fetched by a scalar load.
    Entry: s[36:37] = list pointer, s38 = blocks in the list (> 0), the first min(3, s38) blocks are in
    (or on their way into) the SGPR ring.  Unrolled x6 = lcm(3 SGPR buffers, 2 VGPR data sets).  The wave's
    share of the NEXT tile's LDS-DMA is spread over the rounds (one at the start, one after each block; the
    caller issues the rest) instead of one burst that would stall all 16 waves on the 64 B/clk address path.
    The last three blocks of a list leave the loop for a straight-line tail (one per loop position, out of
    line): lists are short (~5 blocks), so per-block loop bookkeeping is a large share of the issue slots."""
    P1 = p1 if lds else (lambda buf, dset: [])
    P2 = (p2_exact if exact else p2) if fma else (lambda buf, dset: [])
    # the same for the LAST block of a list of four or more blocks: its upper half is skipped when it is all padding
    P1L = (lambda buf, dset: p1_last(buf, dset, 8)) if lds else (lambda buf, dset: [])
    P2L = (lambda buf, dset: p2_last(buf, dset, 8, exact)) if fma else (lambda buf, dset: [])
    o = []
    for _ in range((1 if STAGE == 1 else 0) if STAGE else DMA_AT_START):
        o += dma_hook()
    o += ["s_waitcnt lgkmcnt(0)"]
    o += P1(RING[0], 0)
    o += ["s_waitcnt lgkmcnt(0)",
          "s_cmp_lt_u32 s38, 4", "s_cbranch_scc1 40f",      # short list: the checked tail
          "s_sub_u32 s38, s38, 3",                           # s38 = blocks the loop may take before three are left
          "11:"]
    for k in range(6):
        cur, nxt = RING[k % 3], RING[(k + 1) % 3]
        dc, dn = k % 2, (k + 1) % 2
        off = (k + 3) * 64
        # more than three blocks left -> block r+1 exists and block r+3 is fetched
        o += ["s_sub_u32 s38, s38, 1", f"s_cbranch_scc1 3{k}f"]
        o += P1(nxt, dn) + P2(cur, dc)
        if HOOK_BEFORE_WAIT:
            o += dma_hook()
        o += ["s_waitcnt lgkmcnt(0)", f"s_load_dwordx16 s[{cur}:{cur + 15}], s[36:37], {hex(off)}"]
        if not HOOK_BEFORE_WAIT:
            o += dma_hook()
    o += ["s_add_u32 s36, s36, 0x180", "s_addc_u32 s37, s37, 0", "s_branch 11b"]
    for k in range(6):   # exactly three blocks left, all in the ring: straight line
        cur, nxt, nx2 = RING[k % 3], RING[(k + 1) % 3], RING[(k + 2) % 3]
        dc, dn = k % 2, (k + 1) % 2
        th = [dma_hook() if j < TAIL_HOOKS else [] for j in range(3)]
        o += [f"3{k}:"] + P1(nxt, dn) + P2(cur, dc) + ["s_waitcnt lgkmcnt(0)"] + th[0]
        o += P1L(nx2, dc) + P2(nxt, dn) + ["s_waitcnt lgkmcnt(0)"] + th[1]
        o += P2L(nx2, dc) + th[2] + ["s_branch 12f"]
    # a list of one to three blocks (ring position 0)
    cur, nxt, nx2 = RING
    o += ["40:", "s_cmp_lt_u32 s38, 2", "s_cbranch_scc1 7f"]
    o += P1(nxt, 1)
    o += ["7:"] + P2(cur, 0) + ["s_waitcnt lgkmcnt(0)"] + dma_hook()
    o += ["s_cmp_lt_u32 s38, 2", "s_cbranch_scc1 12f", "s_cmp_lt_u32 s38, 3", "s_cbranch_scc1 7f"]
    o += P1(nx2, 0)
    o += ["7:"] + P2(nxt, 1) + ["s_waitcnt lgkmcnt(0)"] + dma_hook()
    o += ["s_cmp_lt_u32 s38, 3", "s_cbranch_scc1 12f"]
    o += P2(nx2, 0) + dma_hook()
    return o


def list_pointer(first_block_sgpr):
    """s[36:37] = stream base + 64 * first block"""
    f = first_block_sgpr
    return [f"s_lshl_b32 s36, s{f}, 6", f"s_lshr_b32 s37, s{f}, 26",
            "s_add_u32 s36, s36, %[blo]", "s_addc_u32 s37, s37, %[bhi]"]


def request_first_blocks(lo, hi, done, slow):
    """scalar loads of the first min(3, s{hi} - s{lo}) blocks of a list into ring buffers A, B, C (no wait);
    lists shorter than three blocks go through the stub `request_stub`"""
    A, B, C = RING
    return list_pointer(lo) + [
        f"s_sub_u32 s88, s{hi}, s{lo}",
        "s_cmp_lt_u32 s88, 3", f"s_cbranch_scc1 {slow}f",
        f"s_load_dwordx16 s[{A}:{A + 15}], s[36:37], 0x0",
        f"s_load_dwordx16 s[{B}:{B + 15}], s[36:37], 0x40",
        f"s_load_dwordx16 s[{C}:{C + 15}], s[36:37], 0x80",
        f"{done}:"]


def request_stub(done, slow):
    A, B, C = RING
    return [f"{slow}:",
            "s_cmp_eq_u32 s88, 0", f"s_cbranch_scc1 {done}b",
            f"s_load_dwordx16 s[{A}:{A + 15}], s[36:37], 0x0",
            "s_cmp_lt_u32 s88, 2", f"s_cbranch_scc1 {done}b",
            f"s_load_dwordx16 s[{B}:{B + 15}], s[36:37], 0x40",
            f"s_branch {done}b"]


def phases(lds=True, fma=True, exact=False):
    """Tile phases t0 .. te-1 of one wave in ONE asm block, so that SGPR state survives the barrier:
    the first blocks of list t+1 are requested BEFORE the barrier that ends phase t (the scalar path serves
    one 64-byte request per ~20 cycles per CU; 16 waves x 3 requests right after a barrier idle the CU for
    ~1000 cycles).  s90 = t, s91/s92/s93 = first block of lists t, t+1, t+2; s[36:37] = pointer of list t on
    entry to a phase (left there by the request of its first blocks).  v22 (LDS base of the tile being read)
    and s89 (LDS destination of the tile being loaded) toggle between the two buffers once per phase."""
    o = ["s_mov_b32 s39, 0",   # (staged form: no piece is staged)
         "s_mov_b32 s90, %[t0]", "s_mov_b32 s91, %[o0]", "s_mov_b32 s92, %[o1]", "s_mov_b32 s93, %[o2]",
         "s_add_u32 s88, s90, 1", "s_and_b32 s88, s88, 1", "s_lshl_b32 s88, s88, 10", "s_add_u32 s89, s88, %[m0wave]",
         "s_and_b32 s88, s90, 1", "s_lshl_b32 s88, s88, 10", f"v_or_b32 v{BASE}, s88, %[lane8]"]
    o += request_first_blocks(91, 92, 20, 22)
    o += ["1:",
          "s_sub_u32 s38, s92, s91",                       # blocks in this list
          "s_add_u32 s88, s90, 1",                         # next tile: four DMA instructions if it is a full one
          "s_cmp_lt_u32 s88, %[nfull]", f"s_cselect_b64 vcc, {PENDING}, 0",
          "s_cmp_eq_u32 s38, 0", "s_cbranch_scc1 12f"]
    o += list_loop(lds, fma, exact)
    o += ["12:"] + dma_rest()   # the rest of the DMA share
    if STAGE:
        o += ["s_waitcnt lgkmcnt(0)"]   # the tile's pieces are in LDS before the wave arrives at the barrier
    # first blocks of the next list (not at the end of the chunk: the ring must be idle when the asm ends)
    o += ["s_add_u32 s88, s90, 1", "s_cmp_lt_u32 s88, %[te]", "s_cbranch_scc0 21f"]
    o += request_first_blocks(92, 93, 21, 23)
    # o3 = first block of list t+3 (lane min(t+3, ntiles) - obase of the offsets register); the SALU -> v_readlane
    # lane-select hazard (4 wait states) is covered by the address arithmetic of the line touch
    o += ["s_add_u32 s38, s90, 3", "s_min_u32 s38, s38, %[ntiles]", "s_sub_u32 s38, s38, %[obase]",
          "s_lshl_b32 s94, s93, 6", "s_lshr_b32 s95, s93, 26", "s_add_u32 s94, s94, %[blo]", "s_addc_u32 s95, s95, %[bhi]",
          "v_readlane_b32 s88, %[offreg], s38",
          "s_mov_b32 s91, s92", "s_mov_b32 s92, s93",
          f"v_xor_b32 v{BASE}, 0x400, v{BASE}",
          f"s_sub_u32 s89, s89, {hex(KB * 1024)}", "s_xor_b32 s89, s89, 0x400",   # back to the wave's first row pair, other buffer
          # touch the first lines of list t+2 (lane i -> line min(i, L-1), L chosen by the launcher from the mean
          # list length): always ONE instruction; lists are consecutive, so lines past a short list are the next one's
          f"global_load_dword v{TOUCH}, %[toff], s[94:95]"]
    o += ["s_mov_b32 s93, s88"] + ([] if STAGE else [
          "s_waitcnt vmcnt(1)"]) + [  # tile t+1 (this wave's share) has landed; the touch may still fly
          "s_nop 0" if os.environ.get("TL_NO_BARRIER") else "s_barrier",
          "s_add_u32 s90, s90, 1", "s_cmp_lt_u32 s90, %[te]", "s_cbranch_scc1 1b",
          "s_waitcnt lgkmcnt(0)", "s_branch 29f"]
    o += request_stub(20, 22) + request_stub(21, 23) + ["29:"]
    return o


def tile0():
    """the wave's four DMA instructions of tile 0 (buffer 0), advancing the walking pointer"""
    return ["s_mov_b32 s89, %[m0wave]", f"s_mov_b64 vcc, {PENDING}", "13:", "s_cbranch_vccz 14f"] + dma_body() + ["s_branch 13b", "14:"]


def store():
    o = [f"v_mov_b32 v{BASE}, %[lo]", f"v_mov_b32 v{BASE + 1}, %[hi]", "s_mov_b32 s36, 0"]
    for j in range(ROWS):
        o += ["s_cmp_ge_i32 s36, %[n]", "s_cbranch_scc1 9f",
              f"global_store_dwordx2 v[{BASE}:{BASE + 1}], v[{ACC0 + 2 * j}:{ACC0 + 1 + 2 * j}], off nt",
              f"v_lshl_add_u64 v[{BASE}:{BASE + 1}], %[stride], 0, v[{BASE}:{BASE + 1}]",
              "s_add_i32 s36, s36, 1"]
    o.append("9:")
    return o


def store1():
    """the first column of the lane's pair only (float32 layout: the lane that holds the last column of an odd-width result)"""
    o = [f"v_mov_b32 v{BASE}, %[lo]", f"v_mov_b32 v{BASE + 1}, %[hi]", "s_mov_b32 s36, 0"]
    for j in range(ROWS):
        o += ["s_cmp_ge_i32 s36, %[n]", "s_cbranch_scc1 9f",
              f"global_store_dword v[{BASE}:{BASE + 1}], v{ACC0 + 2 * j}, off nt",
              f"v_lshl_add_u64 v[{BASE}:{BASE + 1}], %[stride], 0, v[{BASE}:{BASE + 1}]",
              "s_add_i32 s36, s36, 1"]
    o.append("9:")
    return o


def _store_perm(op, src):
    """the wave's rows through a ROW MAP (round 5, balanced layouts): %[rm] points at the group's ROWS row numbers (int32;
    negative = an unused slot of the group), row j goes to base + row[j] * stride.  The row numbers arrive with three scalar
    loads (ROWS + 1 <= 36 dwords are readable), the address is the lane's base + a 64-bit scalar product."""
    assert ROWS <= 35
    t = BASE + 2
    o = [f"v_mov_b32 v{BASE}, %[lo]", f"v_mov_b32 v{BASE + 1}, %[hi]",
         "s_load_dwordx16 s[40:55], %[rm], 0x0", "s_load_dwordx16 s[56:71], %[rm], 0x40", "s_load_dwordx4 s[72:75], %[rm], 0x80",
         "s_waitcnt lgkmcnt(0)"]
    for j in range(ROWS):
        o += [f"s_cmp_lt_i32 s{40 + j}, 0", f"s_cbranch_scc1 {200 + j}f",
              f"s_mul_hi_u32 s89, s{40 + j}, %[stride]", f"s_mul_i32 s88, s{40 + j}, %[stride]",
              f"v_lshl_add_u64 v[{t}:{t + 1}], s[88:89], 0, v[{BASE}:{BASE + 1}]",
              f"{op} v[{t}:{t + 1}], {src(j)}, off nt", f"{200 + j}:"]
    return o


def store_perm():
    return _store_perm("global_store_dwordx2", lambda j: f"v[{ACC0 + 2 * j}:{ACC0 + 1 + 2 * j}]")


def store1_perm():
    return _store_perm("global_store_dword", lambda j: f"v{ACC0 + 2 * j}")


def zero():
    return [f"v_mov_b32 v{r}, 0" for r in range(JUNK, ACC0 + 2 * ROWS)]


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


def i32_variant():
    """the float32 phases with integer opcodes (always the two-step form: a product, then an add under the index mode)"""
    global I32
    I32 = True
    try:
        return [lit("TL_ASM_PHASES_I32", phases(True, True, True))]
    finally:
        I32 = False


def main():
    out = ["// GENERATED by tools/gen_tiled_asm.py - do not edit.\n",
           f"#define TL_ASM_KB {KB}\n#define TL_ASM_DMA_PER_TILE {DMA_PER_TILE}\n#define TL_ASM_RG {ROWS}\n#define TL_ASM_WAVES {WAVES}\n"
           f"#define TL_ASM_COMP {COMP}\n#define TL_ASM_TOUCH \"v{TOUCH}\"\n#define TL_ASM_BASE \"v{BASE}\"\n",
           lit("TL_ASM_PHASES", phases()),
           lit("TL_ASM_PHASES_EXACT", phases(True, True, True)),
           lit("TL_ASM_PHASES_NOFMA", phases(True, False)),
           lit("TL_ASM_PHASES_NOLDS", phases(False, False)),
           *f64_variants(),
           *i32_variant(),
           lit("TL_ASM_TILE0", tile0()),
           lit("TL_ASM_STORE", store()),
           lit("TL_ASM_STORE1", store1()),
           lit("TL_ASM_STORE_PERM", store_perm()),
           lit("TL_ASM_STORE1_PERM", store1_perm()),
           lit("TL_ASM_ZERO", zero()),
           f"#define TL_CLOB_SGPR {clob('s', 36, 95)}\n",
           f"#define TL_CLOB_TMP {clob('v', WADDR if STAGE else BASE, JUNK - 1)}\n",
           f"#define TL_CLOB_ACC {clob('v', JUNK, ACC0 + 2 * ROWS - 1)}\n"]
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sparse_amd", "csrc", "spmm_tiled_asm.inc")
    with open(p, "w") as f:
        f.write("\n".join(out))


if __name__ == "__main__":
    main()
