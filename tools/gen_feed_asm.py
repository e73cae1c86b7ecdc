#!/usr/bin/env python3
"""Generates garlic_amd/csrc/feed_loop_gfx950.inc: the hand-scheduled interior loop of lod_feed_kernel
(feed_kernel.hpp) -- the thinned LOD scores of the KDE feed, every wavefront a chain of its own (gfx950, wave64).

Why by hand: a wavefront alone issues one instruction per ~4 cycles whatever the instruction is (tools/ubench/
issue_rates.hip), and the length of the longest run of windows x the pace of ONE wave is the kernel's critical
path.  hipcc's version of the same loop spends 12.3 instructions per window (2.7 of them scalar, 0.8 branches)
and 37 % of its cycles in s_waitcnt (rocprofv3 SQ counters, tools/exp/feed_pmc.sh): 84 cycles per window.
Here a window costs 8.1 instructions and every wait is counted.

One block for all four waves of a workgroup (no roles: each wave owns one 64-individual block of the same run of
windows; lane = individual).  Per window and lane:
    2 x  byte extract      the genotype's term offset (genotype * 8) of the entering and of the leaving SNP
    2 x  ds_read_b64       its term from the SNP's row {lod(0), lod(1), lod(2), +0.0} in LDS (immediate offset)
    2 x  v_add_f64         acc = (acc - t_out) + t_in, two roundings as src/garlic-roh.cpp:98-100
The genotype offsets come from the packed 2-bit words without a table: per 16 SNPs four masked copies
X_c = (word shifted by 3 - 2c) & 0x18181818, byte m of X_c = 8 * genotype of SNP 4m + c (8 instructions per
word instead of a shift and a mask per SNP).

Tiles of 32 windows; the loop is unrolled over the 4 slots of the two term-row rings so that every LDS address
is an immediate:
    TRAIL ring  slot t % 4  <- rows of the leaving SNPs of tile t   (1 KB = 32 rows)
    LEAD  ring  slot t % 4  <- rows of the entering SNPs of tile t
Tile t, every wave:  s_barrier (all waves are done with tile t-1 and with their look-ahead into tile t)
                   | LDS-DMA of its quarter (256 B) of both chunks of tile t+3
                   | 4 genotype words (both streams) of tile t+2 into the word rings (8 registers per stream)
                   | 8 groups of 4 windows, software-pipelined: the look-ups of group g+1 are issued around the
                     adds of group g (at most 12 LDS reads in flight; the LGKM counter holds 15); group 7 runs
                     beside the first look-ups of tile t+1
                   | s_waitcnt vmcnt(6): everything requested during tile t-1 has landed
A batch of 8 windows keeps its 8 accumulators (the chain writes each into a register pair of its own); a
sampled locus among them (thinned output: every step-th locus, src/garlic-data.cpp:2036) is picked with
s_set_gpr_idx and stored from the lanes, 8 B each.

The block runs 4 n tiles starting at a tile index that is a multiple of 4, all of them interior (every window
of the tile is a rolling update of the run); lod_feed_kernel runs the other tiles through its compiler-generated
path, which keeps the same ring protocol.  A wave without a block (`active` = 0) only keeps the protocol.

Environment hook for experiments: GARLIC_FEED_ABLATE (nodp, nolds, nodma, nowords, nobar, nocapture).
"""
import os

ABL = os.environ.get("GARLIC_FEED_ABLATE", "")

# ---- LDS map (bytes, workgroup-relative; feed_kernel.hpp takes GARLIC_FEED_LDS_* from the generated file)
LDS_MISC = 0            # item word etc. (compiler-generated code only)
TRAIL_BASE = 1024
LEAD_BASE = TRAIL_BASE + 4 * 1024
LDS_TOTAL = LEAD_BASE + 4 * 1024
NSLOT = 4
AHEAD = 3               # chunks of tile t + AHEAD are requested during tile t

# ---- fixed VGPRs (clobbered by the block)
V_WL = 20               # lead word ring, 8 registers: word i of the stream (from the loop's first tile) at i % 8
V_WT = 28               # trail word ring
V_XL = 36               # lead:  LO_A c0..3 | LO_B c0..3 | HI c0..3   (tiles alternate between LO_A and LO_B)
V_XT = 48               # trail: the same
V_ADDR = 60             # 8 look-up offsets
V_BUF = [68, 84]        # two term buffers: 4 steps x {t_in, t_out} (16 registers each)
V_ACC = 100             # 8 accumulators of the current batch (16 registers)
V_LANE4 = 116           # lane * 4: genotype word offset inside a word row
V_DMAOFF = 117          # lane * 4 + wave * 256: this wave's quarter of a chunk
V_STOFF = 118           # the lane's row * row pitch (bytes) of the sample matrix
V_S = 120               # the sampled accumulator (2 registers)
V_FUN = 122             # funnel-shifted genotype words being spread (2 registers: entering, leaving)
CLOBBER_V = list(range(20, 124))
# ---- fixed SGPRs
S_PLW, S_PTW = 40, 42   # genotype word streams: address of word row 0 of the loop's first tile (+ lane * 4)
S_PTL, S_PTT = 44, 46   # term-row chunks: address of the chunk of tile 0 of the loop (lead / trail)
S_OUT = 48              # sample matrix: address of the next sample's column in the block's first row
S_NEXT = 50             # windows from the current batch's first window to the next sampled locus
S_STEP = 51
S_CNT = 52              # iterations (4 tiles each) left
S_SHL, S_SHT = 53, 54   # funnel shifts of the two streams
S_W256 = 55             # wave * 256
S_TMP = 56
S_IDX = 57
S_MASK, S_EXEC = 58, 60   # lanes that have a row in the sample matrix; saved exec
S_CUT = 62              # bits variant: the LOD cutoff (2 registers)
CLOBBER_S = list(range(40, 64))
# bits variant (GARLIC_FEED_BITS_LOOP_ASM): instead of sampled scores the loop leaves ONE BIT per window and lane --
# score >= cutoff -- 32 of them per tile in V_S, stored as one dword per lane and tile; V_S + 1 holds zero, the
# window's bit goes from VCC into the dword by an add-with-carry
V_BITS, V_ZERO = V_S, V_S + 1

MASK = "0x18181818"


class Gen:
    """instruction list + a model of the in-order LGKM counter (LDS reads only)"""

    def __init__(self, bits=False):
        self.out = []
        self.issued = 0
        self.complete = 0
        self.bits = bits

    def emit(self, s):
        self.out.append(s)

    def lds(self, s):
        self.out.append(s)
        self.issued += 1
        return self.issued

    def wait_lds(self, op):
        if op <= self.complete:
            return
        n = self.issued - op
        assert n <= 15
        self.emit(f"s_waitcnt lgkmcnt({n})")
        self.complete = op


def pair(r):
    return f"v[{r}:{r + 1}]"


def word(ring, i):
    return ring + i % 8


def xreg(base, which, c):
    """which: 'A' / 'B' (the LO copies of even / odd tiles) or 'H'"""
    return base + {"A": 0, "B": 4, "H": 8}[which] + c


def gen_copies(e, base, which, src):
    """X_c = (src shifted by 3 - 2c) & 0x18181818: byte m of X_c = 8 * genotype of step 4m + c"""
    for c, op in enumerate(("v_lshlrev_b32_e32 v{d}, 3, v{s}", "v_lshlrev_b32_e32 v{d}, 1, v{s}",
                            "v_lshrrev_b32_e32 v{d}, 1, v{s}", "v_lshrrev_b32_e32 v{d}, 3, v{s}")):
        d = xreg(base, which, c)
        e(op.format(d=d, s=src))
        e(f"v_and_b32_e32 v{d}, {MASK}, v{d}")


def extract(e, dst, src, m):
    if m == 0:
        e(f"v_and_b32_e32 v{dst}, 0xff, v{src}")
    elif m == 3:
        e(f"v_lshrrev_b32_e32 v{dst}, 24, v{src}")
    else:
        e(f"v_bfe_u32 v{dst}, v{src}, {8 * m}, 8")


def group_source(u, g):
    """X registers and LDS slot of group g (0..7: this tile; 8: group 0 of the next tile) of unrolled tile u"""
    lo_this = "A" if u % 2 == 0 else "B"
    lo_next = "B" if u % 2 == 0 else "A"
    if g == 8:
        return lo_next, 0, (u + 1) % NSLOT, 0
    return (lo_this if g < 4 else "H"), g % 4, u % NSLOT, 4 * g


def gen_E(g_, u, g):
    """the 8 look-up offsets of group g: steps 4g .. 4g+3, entering stream in ADDR[0..3], leaving in ADDR[4..7]"""
    which, m, _, _ = group_source(u, g)
    for c in range(4):
        extract(g_.emit, V_ADDR + c, xreg(V_XL, which, c), m)
    for c in range(4):
        extract(g_.emit, V_ADDR + 4 + c, xreg(V_XT, which, c), m)


def gen_R(g_, u, g, steps):
    """look-ups of `steps` (subset of 0..3) of group g into buffer g % 2"""
    _, _, slot, j0 = group_source(u, g)
    buf = V_BUF[g % 2]
    last = 0
    for i in steps:
        j = j0 + i
        if "nolds" in ABL:
            continue
        g_.lds(f"ds_read_b64 {pair(buf + 4 * i)}, v{V_ADDR + i} offset:{LEAD_BASE + slot * 1024 + 32 * j}")
        last = g_.lds(f"ds_read_b64 {pair(buf + 4 * i + 2)}, v{V_ADDR + 4 + i} offset:{TRAIL_BASE + slot * 1024 + 32 * j}")
    return last


def gen_C(g_, g, steps):
    """the chain through `steps` of group g: accumulator of window 4(g%2) + i of the batch into its own pair"""
    buf = V_BUF[g % 2]
    for i in steps:
        j = 4 * (g % 2) + i
        dst = V_ACC + 2 * j
        prev = V_ACC + 2 * ((j - 1) % 8)
        if "nodp" in ABL:
            continue
        g_.emit(f"v_add_f64 {pair(dst)}, {pair(prev)}, -{pair(buf + 4 * i + 2)}")
        g_.emit(f"v_add_f64 {pair(dst)}, {pair(dst)}, {pair(buf + 4 * i)}")


def gen_capture(g_, u, b):
    """out of line: the sampled accumulator(s) of batch b of tile u.  S_NEXT has gone below zero (mod 2^32)."""
    e = g_.emit
    e(f"CAP_{u}_{b}_%=:")
    e(f"s_add_u32 s{S_IDX}, s{S_NEXT}, 8")                  # window of the batch
    e(f"s_lshl_b32 s{S_IDX}, s{S_IDX}, 1")                  # its register pair
    e(f"s_set_gpr_idx_on s{S_IDX}, 0x1")                    # SRC0 relative
    e(f"v_mov_b32_e32 v{V_S}, v{V_ACC}")
    e(f"v_mov_b32_e32 v{V_S + 1}, v{V_ACC + 1}")
    e("s_set_gpr_idx_off")
    e(f"s_mov_b64 exec, s[{S_MASK}:{S_MASK + 1}]")          # individuals without a row: not stored
    e(f"global_store_dwordx2 v{V_STOFF}, {pair(V_S)}, s[{S_OUT}:{S_OUT + 1}]")
    e(f"s_mov_b64 exec, s[{S_EXEC}:{S_EXEC + 1}]")
    e(f"s_add_u32 s{S_OUT}, s{S_OUT}, 8")
    e(f"s_addc_u32 s{S_OUT + 1}, s{S_OUT + 1}, 0")
    e(f"s_add_i32 s{S_NEXT}, s{S_NEXT}, s{S_STEP}")
    e(f"s_cmp_lt_i32 s{S_NEXT}, 0")
    e(f"s_cbranch_scc1 CAP_{u}_{b}_%=")
    e(f"s_branch CAPRET_{u}_{b}_%=")


def gen_bits(g_, b):
    """bits variant, batch b (windows 8b .. 8b+7 of the tile, accumulators in V_ACC): the compare (cutoff <= score; false
    for a NaN) leaves the window's bit in VCC, and an add-with-carry shifts it into the tile's dword,
    bits = 2 * bits + carry: two instructions per window.  The first window comes out on top: gen_tile reverses the
    dword (v_bfrev_b32) before it is stored."""
    e = g_.emit
    for j in range(8):
        e(f"v_cmp_le_f64_e32 vcc, s[{S_CUT}:{S_CUT + 1}], {pair(V_ACC + 2 * j)}")
        src = V_ZERO if (b == 0 and j == 0) else V_BITS      # the tile's first window starts the dword
        e(f"v_addc_co_u32_e32 v{V_BITS}, vcc, v{src}, v{src}, vcc")


def gen_tile(g_, u):
    """tile t = 4 i + u of the loop"""
    e = g_.emit
    lo_next = "B" if u % 2 == 0 else "A"
    if "nobar" not in ABL:
        e("s_barrier")
    # ---- requests: this wave's quarter of the two chunks of tile t + 3, the genotype words 7 and 8 of the tile (tile
    #      t + 2 funnel-shifts word 7 for its look-ahead).  The instruction between an M0 write and the LDS-DMA that
    #      reads it is the wait state M0 needs.
    slot = (u + AHEAD) % NSLOT
    fun_l = f"v_alignbit_b32 v{V_FUN}, v{word(V_WL, 2 * u + 2)}, v{word(V_WL, 2 * u + 1)}, s{S_SHL}"
    fun_t = f"v_alignbit_b32 v{V_FUN + 1}, v{word(V_WT, 2 * u + 2)}, v{word(V_WT, 2 * u + 1)}, s{S_SHT}"
    if "nodma" not in ABL:
        e(f"s_add_u32 m0, s{S_W256}, {LEAD_BASE + slot * 1024}")
        e(fun_l)
        e(f"global_load_lds_dword v{V_DMAOFF}, s[{S_PTL}:{S_PTL + 1}]")
        e(f"s_add_u32 m0, s{S_W256}, {TRAIL_BASE + slot * 1024}")
        e(fun_t)
        e(f"global_load_lds_dword v{V_DMAOFF}, s[{S_PTT}:{S_PTT + 1}]")
        bump(e, S_PTL, 1024)
        bump(e, S_PTT, 1024)
    else:
        e(fun_l)
        e(fun_t)
    if "nowords" not in ABL:
        for ring, ptr in ((V_WL, S_PLW), (V_WT, S_PTW)):
            for i in (7, 8):
                e(f"global_load_dword v{word(ring, 2 * u + i)}, v{V_LANE4}, s[{ptr}:{ptr + 1}] offset:{(2 * u + i) * 256 - 2048}")
    # ---- the tile's second word (steps 16..31) of both streams: masked copies
    gen_copies(e, V_XL, "H", V_FUN)
    gen_copies(e, V_XT, "H", V_FUN + 1)

    # ---- 8 groups; on entry the look-ups of group 0 are in flight (issued beside group 7 of the previous tile)
    g_.issued = 8
    g_.complete = 0
    r_prev = 8                                   # id of the last look-up of R(0)
    for g in range(8):
        if g == 6:
            # first word of the next tile: funnel shift + copies (its words were waited for at the end of tile t-1)
            e(f"v_alignbit_b32 v{V_FUN}, v{word(V_WL, 2 * u + 3)}, v{word(V_WL, 2 * u + 2)}, s{S_SHL}")
            e(f"v_alignbit_b32 v{V_FUN + 1}, v{word(V_WT, 2 * u + 3)}, v{word(V_WT, 2 * u + 2)}, s{S_SHT}")
            gen_copies(e, V_XL, lo_next, V_FUN)
            gen_copies(e, V_XT, lo_next, V_FUN + 1)
        gen_E(g_, u, g + 1)
        gen_R(g_, u, g + 1, (0, 1))
        g_.wait_lds(r_prev)                      # R(g) complete (the look-ups just issued stay in flight)
        gen_C(g_, g, (0, 1))
        r_prev = gen_R(g_, u, g + 1, (2, 3))
        gen_C(g_, g, (2, 3))
        if g % 2 == 1 and g_.bits:
            gen_bits(g_, g // 2)
        elif g % 2 == 1 and "nocapture" not in ABL:
            b = g // 2
            e(f"s_sub_u32 s{S_NEXT}, s{S_NEXT}, 8")          # borrow: a sampled locus among the batch's 8 windows
            e(f"s_cbranch_scc1 CAP_{u}_{b}_%=")
            e(f"CAPRET_{u}_{b}_%=:")
    # ---- everything requested during tile t-1 has landed: genotype words of tile t+1, chunks of tile t+2
    nreq = (0 if "nodma" in ABL else 2) + (0 if "nowords" in ABL else 4)
    if g_.bits:
        # the tile's 32 bits: one dword per lane (individuals without a row are masked out); it stays in flight too
        e(f"v_bfrev_b32_e32 v{V_BITS}, v{V_BITS}")
        e(f"s_mov_b64 exec, s[{S_MASK}:{S_MASK + 1}]")
        e(f"global_store_dword v{V_STOFF}, v{V_BITS}, s[{S_OUT}:{S_OUT + 1}]")
        e(f"s_mov_b64 exec, s[{S_EXEC}:{S_EXEC + 1}]")
        e(f"s_add_u32 s{S_OUT}, s{S_OUT}, 4")
        e(f"s_addc_u32 s{S_OUT + 1}, s{S_OUT + 1}, 0")
        nreq += 1
    e(f"s_waitcnt vmcnt({nreq})")


def gen_idle_tile(g_, u):
    """a wave without a block: the barrier and its quarter of the chunks"""
    e = g_.emit
    if "nobar" not in ABL:
        e("s_barrier")
    slot = (u + AHEAD) % NSLOT
    if "nodma" not in ABL:
        e(f"s_add_u32 m0, s{S_W256}, {LEAD_BASE + slot * 1024}")
        e("s_nop 0")
        e(f"global_load_lds_dword v{V_DMAOFF}, s[{S_PTL}:{S_PTL + 1}]")
        e(f"s_add_u32 m0, s{S_W256}, {TRAIL_BASE + slot * 1024}")
        e("s_nop 0")
        e(f"global_load_lds_dword v{V_DMAOFF}, s[{S_PTT}:{S_PTT + 1}]")
        bump(e, S_PTL, 1024)
        bump(e, S_PTT, 1024)
        e("s_waitcnt vmcnt(2)")


def bump(e, ptr, n):
    e(f"s_add_u32 s{ptr}, s{ptr}, {n}")
    e(f"s_addc_u32 s{ptr + 1}, s{ptr + 1}, 0")


def gen_all(bits=False):
    g_ = Gen(bits)
    e = g_.emit
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e(f"s_mov_b64 s[{S_PTL}:{S_PTL + 1}], %[ptl]")
    e(f"s_mov_b64 s[{S_PTT}:{S_PTT + 1}], %[ptt]")
    e(f"s_mov_b32 s{S_CNT}, %[niter]")
    e(f"s_lshl_b32 s{S_W256}, %[wave], 8")
    e(f"v_lshlrev_b32_e32 v{V_LANE4}, 2, %[lane]")
    e(f"v_add_u32_e32 v{V_DMAOFF}, s{S_W256}, v{V_LANE4}")
    # chunks: the block requests tiles 3, 4, .. (0..2 are in the rings)
    bump(e, S_PTL, AHEAD * 1024)
    bump(e, S_PTT, AHEAD * 1024)
    e("s_cmp_eq_u32 %[active], 0")
    e("s_cbranch_scc1 IDLE_%=")
    e(f"s_mov_b64 s[{S_PLW}:{S_PLW + 1}], %[plw]")
    e(f"s_mov_b64 s[{S_PTW}:{S_PTW + 1}], %[ptw]")
    e(f"s_mov_b64 s[{S_OUT}:{S_OUT + 1}], %[out]")
    e(f"s_mov_b32 s{S_NEXT}, %[next]")
    e(f"s_mov_b32 s{S_STEP}, %[step]")
    e(f"s_mov_b32 s{S_SHL}, %[shl]")
    e(f"s_mov_b32 s{S_SHT}, %[sht]")
    e(f"s_mov_b64 s[{S_EXEC}:{S_EXEC + 1}], exec")
    e(f"v_cmp_le_i32_e64 s[{S_MASK}:{S_MASK + 1}], 0, %[row]")     # lanes with a row in the sample matrix
    e(f"v_mul_lo_u32 v{V_STOFF}, %[row], %[pitch8]")
    if bits:
        e(f"s_mov_b64 s[{S_CUT}:{S_CUT + 1}], %[cut]")
        e(f"v_mov_b32_e32 v{V_ZERO}, 0")
    e(f"v_mov_b64 {pair(V_ACC + 14)}, %[acc]")
    # words 0..6 of both streams (tile t funnel-shifts words 2t .. 2t + 3); the loop loads from word 7 on
    for ring, ptr in ((V_WL, S_PLW), (V_WT, S_PTW)):
        for i in range(7):
            e(f"global_load_dword v{word(ring, i)}, v{V_LANE4}, s[{ptr}:{ptr + 1}] offset:{i * 256}")
    bump(e, S_PLW, 2048)       # the loop's word offsets are (2u + i) * 256 - 2048
    bump(e, S_PTW, 2048)
    e("s_waitcnt vmcnt(0)")
    # first word of tile 0 -> LO_A, and its group 0 look-ups
    e(f"v_alignbit_b32 v{V_FUN}, v{word(V_WL, 1)}, v{word(V_WL, 0)}, s{S_SHL}")
    e(f"v_alignbit_b32 v{V_FUN + 1}, v{word(V_WT, 1)}, v{word(V_WT, 0)}, s{S_SHT}")
    gen_copies(e, V_XL, "A", V_FUN)
    gen_copies(e, V_XT, "A", V_FUN + 1)
    # group 0 of tile 0 = "group 8" of a tile u = 3 (next LO = A, next slot = 0)
    gen_E(g_, 3, 8)
    gen_R(g_, 3, 8, (0, 1, 2, 3))
    e("LOOP_%=:")
    for u in range(4):
        gen_tile(g_, u)
    bump(e, S_PLW, 2048)
    bump(e, S_PTW, 2048)
    e(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    e(f"s_cmp_lg_u32 s{S_CNT}, 0")
    e("s_cbranch_scc1 LOOP_%=")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e(f"v_mov_b64 %[acc], {pair(V_ACC + 14)}")
    e(f"s_mov_b32 %[next_out], s{S_NEXT}")
    e(f"s_mov_b64 %[out_out], s[{S_OUT}:{S_OUT + 1}]")
    e("s_branch DONE_%=")
    if "nocapture" not in ABL and not bits:
        for u in range(4):
            for b in range(4):
                gen_capture(g_, u, b)
    e("IDLE_%=:")
    e("IDLE_LOOP_%=:")
    for u in range(4):
        gen_idle_tile(g_, u)
    e(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    e(f"s_cmp_lg_u32 s{S_CNT}, 0")
    e("s_cbranch_scc1 IDLE_LOOP_%=")
    e("s_waitcnt vmcnt(0)")
    e("DONE_%=:")
    return g_.out


def main():
    lines = gen_all()
    bits_lines = gen_all(bits=True)
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(os.environ.get("GARLIC_GEN_OUT") or os.path.join(here, "..", "garlic_amd", "csrc"), "feed_loop_gfx950.inc")
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_feed_asm.py -- do not edit; see that file for the schedule.\n")
        f.write("// One inline-asm block: interior tiles of lod_feed_kernel, every wave a chain of its own (gfx950).\n")
        f.write(f"#define GARLIC_FEED_LDS_TOTAL {LDS_TOTAL}\n")
        f.write(f"#define GARLIC_FEED_LDS_TRAIL {TRAIL_BASE}\n")
        f.write(f"#define GARLIC_FEED_LDS_LEAD {LEAD_BASE}\n")
        f.write(f"#define GARLIC_FEED_AHEAD {AHEAD}\n")
        f.write("#define GARLIC_FEED_LOOP_ASM \\\n")
        for ln in lines:
            f.write('    "%s\\n\\t" \\\n' % ln)
        f.write('    ""\n')
        f.write("// the same loop leaving one bit per window and lane (score >= cutoff) instead of sampled scores\n")
        f.write("#define GARLIC_FEED_BITS_LOOP_ASM \\\n")
        for ln in bits_lines:
            f.write('    "%s\\n\\t" \\\n' % ln)
        f.write('    ""\n')
        f.write("#define GARLIC_FEED_LOOP_CLOBBERS \\\n    ")
        regs = ['"v%d"' % r for r in CLOBBER_V] + ['"s%d"' % r for r in CLOBBER_S]
        regs += ['"memory"', '"scc"', '"vcc"', '"m0"']
        chunks = [", ".join(regs[i:i + 12]) for i in range(0, len(regs), 12)]
        f.write(", \\\n    ".join(chunks) + "\n")
    n_instr = sum(1 for ln in lines if not ln.endswith(":"))
    print(f"wrote {os.path.normpath(path)}: {n_instr} instructions, LDS {LDS_TOTAL} B")


if __name__ == "__main__":
    main()
