#!/usr/bin/env python3
"""Generates garlic_amd/csrc/feed_loop_gfx950.inc: the hand-scheduled interior loop of lod_feed_kernel /
lod_bits_kernel (feed_kernel.hpp) -- the window recurrence of src/garlic-roh.cpp:92-100 with every wavefront a
chain of its own (gfx950, wave64).

Why by hand: a wavefront alone issues one instruction per ~4 cycles whatever the instruction is (tools/ubench/
issue_rates.hip), and the length of the longest run of windows x the pace of ONE wave is the kernel's critical
path (10M SNPs: 440k windows in one run).  Round 3's loop spent 8.1 instructions per window (10.1 with the
coverage bit); this one 5.2 (7.2):

  per window and lane (lane = individual)
    1 x  SDWA nibble extract   the LDS offset 16 * (4 * g_out + g_in) of the window's genotype pair: the packed
                               2-bit words of the entering and the leaving stream are merged once per 16 SNPs into
                               two words of 4-bit codes (even / odd SNPs: v_lshl + v_bfi each), and an SDWA byte
                               select with an AND / a shift into byte 0 turns a nibble into the offset in ONE
                               instruction (round 3: four masked copies per stream and a byte extract per stream
                               and window)
    1 x  ds_read_b128          {t_out[g_out], t_in[g_in]} from the window's 16-entry pair table (round 3: two
                               ds_read_b64 from two rows)
    2 x  v_add_f64             acc = (acc - t_out) + t_in, two roundings as src/garlic-roh.cpp:98-100
   (2 x  v_cmp_le_f64 + v_addc_co_u32: the coverage bit, bits variant)

  pair tables ("comb ring")    per tile of 32 windows 32 x 16 x 16 B = 8 KB, four slots (tile t in slot t % 4), shared
                               by the workgroup's four waves; every wave builds a quarter of the table of tile t + 2
                               during tile t: the term rows {lod(0), lod(1), lod(2), +0.0} of the entering and leaving
                               SNPs straight from the term table in memory (4 global_load_dwordx2 a tile and wave,
                               issued a tile and a quarter ahead: no LDS-DMA, no M0, no raw-row rings), then two
                               ds_write_b128.  One s_barrier per tile is the ring's only protocol:
                               tile t reads slots t, t + 1 (look-ahead), writes slot t + 2; slot t + 2 = t - 2 was last
                               read before barrier t - 1.
  genotype words               the lane's own packed words of both streams straight into registers (8-word rings)
  samples (feed)               a sampled locus (every step-th, src/garlic-data.cpp:2036) is picked from the batch's 8
                               accumulators with s_set_gpr_idx and kept; four of them leave as one 32-byte piece per
                               lane (round 3: 8 B each -- 4.6 x the bytes at the memory controller)
  bits                         a tile's 32 bits are a dword per lane; eight tiles' dwords leave as one aligned 32-byte
                               piece per lane (round 3: 4 B each)

The block runs 8 n tiles, all of them interior (every window a rolling update); lod_feed_kernel runs the other tiles
through its compiler-generated path, which needs no ring: it reads the term table itself.  A wave without a block
(`active` = 0) only builds its quarter of the tables.

Environment hook for experiments: GARLIC_FEED_ABLATE (nodp, nolds, nobar, nocapture, nowords, nocomb).
"""
import os

ABL = os.environ.get("GARLIC_FEED_ABLATE", "")

# ---- LDS map (bytes, workgroup-relative; feed_kernel.hpp takes GARLIC_FEED_LDS_* from the generated file)
LDS_MISC = 0            # item word etc. (compiler-generated code only)
COMB_BASE = 1024
COMB_SLOT = 8192        # 32 windows x 16 pairs x 16 B
NSLOT = 4
LDS_TOTAL = COMB_BASE + NSLOT * COMB_SLOT
UNROLL = 8              # tiles per loop iteration
TERM_AHEAD = 3          # the term rows of tile t + 3 are requested at the start of tile t, written at group 2 of tile t + 1

# ---- fixed VGPRs (clobbered by the block)
V_WL = 20               # lead word ring, 8 registers: word i of the stream (from the loop's first tile) at i % 8
V_WT = 28               # trail word ring
V_H = 36                # 4-bit codes 4 g_out + g_in: LO_A {even, odd} | LO_B {even, odd} | HI {even, odd}
V_FUN = 42              # funnel-shifted genotype words (2 registers: entering, leaving)
V_BUF = [44, 60]        # two look-up buffers: 4 windows x {t_out, t_in} (16 registers each); a window's offset is extracted into its first register
V_ACC = 76              # 8 accumulators of the current batch (16 registers)
V_TB = [92, 100]        # term rows on their way into the pair tables: 2 passes x {t_out, t_in} (8 registers each), tiles alternate
V_LANE4 = 108           # lane * 4: genotype word offset inside a word row
V_CT = 109              # this lane's leaving-SNP term:  32 * (8 wave + lane / 16) + 8 * ((lane % 16) / 4)
V_CL = 110              # this lane's entering-SNP term: 32 * (8 wave + lane / 16) + 8 * (lane % 4)
V_CW = 111              # its pair in the table:         256 * (8 wave + lane / 16) + 16 * (lane % 16)
V_STOFF = 112           # the lane's row * row pitch (bytes) of the sample / bit matrix
V_S = 114               # the sampled accumulator (2 registers)
V_SR = 116              # feed: four kept samples (8 registers); bits: the dwords of the iteration's 8 tiles
CLOBBER_V = list(range(20, 124))
# ---- fixed SGPRs
S_PLW, S_PTW = 40, 42   # genotype word streams: address of word row 0 of the loop's first tile (+ lane * 4)
S_PTL, S_PTT = 44, 46   # term rows of the entering / leaving SNPs, TERM_AHEAD tiles ahead of the iteration's first tile
S_OUT = 48              # sample matrix: address of the next piece's column in the block's first row
S_NEXT = 50             # windows from the current batch's first window to the next sampled locus
S_STEP = 51
S_CNT = 52              # iterations left
S_SHL, S_SHT = 53, 54   # funnel shifts of the two streams
S_F0 = 55               # 0xf0
S_TMP = 56
S_IDX = 57
S_MASK, S_EXEC = 58, 60   # lanes that have a row in the sample matrix; saved exec
S_CUT = 62              # bits variant: the LOD cutoff (2 registers)
S_MCC = 64              # 0xcccccccc
S_NCAP = 65             # feed: samples kept (0..3)
CLOBBER_S = list(range(40, 66))
# bits variant (GARLIC_FEED_BITS_LOOP_ASM): instead of sampled scores the loop leaves ONE BIT per window and lane --
# score >= cutoff -- 32 of them per tile in V_BITS; V_ZERO holds zero, the window's bit goes from VCC into the dword
# by an add-with-carry
V_BITS, V_ZERO = V_S, V_S + 1


class Gen:
    """instruction list + a model of the in-order LGKM counter (LDS operations only)"""

    def __init__(self, bits=False):
        self.out = []
        self.issued = 0
        self.complete = 0
        self.bits = bits
        self.last_sdwa_dst = None

    def emit(self, s):
        self.out.append(s)
        self.last_sdwa_dst = None

    def sdwa(self, s, dst):
        self.out.append(s)
        self.last_sdwa_dst = dst

    def lds(self, s, addr=None):
        # (an SDWA result needs an instruction between its write and its reader on gfx940+)
        assert addr is None or addr != self.last_sdwa_dst, s
        self.out.append(s)
        self.last_sdwa_dst = None
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


def quad(r):
    return f"v[{r}:{r + 3}]"


def word(ring, i):
    return ring + i % 8


def hreg(which, odd):
    """which: 'A' / 'B' (the LO words of even / odd tiles) or 'H'"""
    return V_H + {"A": 0, "B": 2, "H": 4}[which] + odd


def gen_codes(e, which):
    """the two code words of 16 SNPs from the funnel-shifted words V_FUN (entering) and V_FUN + 1 (leaving):
    nibble i of the even word = 4 g_out + g_in of SNP 2 i, of the odd word of SNP 2 i + 1"""
    he, ho = hreg(which, 0), hreg(which, 1)
    e(f"v_lshlrev_b32_e32 v{he}, 2, v{V_FUN + 1}")
    e(f"v_lshrrev_b32_e32 v{ho}, 2, v{V_FUN}")
    e(f"v_bfi_b32 v{he}, s{S_MCC}, v{he}, v{V_FUN}")          # (mask & out << 2) | (~mask & in)
    e(f"v_bfi_b32 v{ho}, s{S_MCC}, v{V_FUN + 1}, v{ho}")      # (mask & out) | (~mask & in >> 2)


def group_source(u, g):
    """code words, byte, LDS slot and first window of group g (0..7: this tile; 8: group 0 of the next tile) of
    unrolled tile u"""
    lo_this = "A" if u % 2 == 0 else "B"
    lo_next = "B" if u % 2 == 0 else "A"
    if g == 8:
        return lo_next, 0, (u + 1) % NSLOT, 0
    return (lo_this if g < 4 else "H"), g % 4, u % NSLOT, 4 * g


def gen_E(g_, u, g):
    """the 4 look-up offsets of group g (windows 4g .. 4g+3 of the tile) into the first register of each window's
    buffer: window i of the group is SNP 4 (g % 4) + i of its word: byte g % 4 of the even (i = 0, 2) / odd word,
    low (i < 2) / high nibble"""
    which, b, _, _ = group_source(u, g)
    buf = V_BUF[g % 2]
    for i in range(4):
        src = hreg(which, i & 1)
        dst = buf + 4 * i
        if i < 2:
            g_.sdwa(f"v_lshlrev_b32_sdwa v{dst}, 4, v{src} dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_{b}", dst)
        else:
            g_.sdwa(f"v_and_b32_sdwa v{dst}, s{S_F0}, v{src} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_{b}", dst)


def gen_R(g_, u, g, steps):
    """look-ups of `steps` (subset of 0..3) of group g into buffer g % 2"""
    _, _, slot, j0 = group_source(u, g)
    buf = V_BUF[g % 2]
    last = 0
    for i in steps:
        if "nolds" in ABL:
            continue
        r = buf + 4 * i
        last = g_.lds(f"ds_read_b128 {quad(r)}, v{r} offset:{COMB_BASE + slot * COMB_SLOT + 256 * (j0 + i)}", r)
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
        g_.emit(f"v_add_f64 {pair(dst)}, {pair(prev)}, -{pair(buf + 4 * i)}")
        g_.emit(f"v_add_f64 {pair(dst)}, {pair(dst)}, {pair(buf + 4 * i + 2)}")


def masked(e, lines):
    e(f"s_mov_b64 exec, s[{S_MASK}:{S_MASK + 1}]")          # individuals without a row: not stored
    for ln in lines:
        e(ln)
    e(f"s_mov_b64 exec, s[{S_EXEC}:{S_EXEC + 1}]")


def gen_capture(g_, u, b):
    """out of line: the sampled accumulator(s) of batch b of tile u.  S_NEXT has gone below zero (mod 2^32).  The
    sample joins the kept ones; four leave together as 32 bytes per lane."""
    e = g_.emit
    e(f"CAP_{u}_{b}_%=:")
    e(f"s_add_u32 s{S_IDX}, s{S_NEXT}, 8")                  # window of the batch
    e(f"s_lshl_b32 s{S_IDX}, s{S_IDX}, 1")                  # its register pair
    e(f"s_set_gpr_idx_on s{S_IDX}, 0x1")                    # SRC0 relative
    e(f"v_mov_b32_e32 v{V_S}, v{V_ACC}")
    e(f"v_mov_b32_e32 v{V_S + 1}, v{V_ACC + 1}")
    e("s_set_gpr_idx_off")
    e(f"s_lshl_b32 s{S_IDX}, s{S_NCAP}, 1")
    e(f"s_set_gpr_idx_on s{S_IDX}, 0x8")                    # DST relative
    e(f"v_mov_b32_e32 v{V_SR}, v{V_S}")
    e(f"v_mov_b32_e32 v{V_SR + 1}, v{V_S + 1}")
    e("s_set_gpr_idx_off")
    e(f"s_add_u32 s{S_NCAP}, s{S_NCAP}, 1")
    e(f"s_cmp_lg_u32 s{S_NCAP}, 4")
    e(f"s_cbranch_scc1 CAPK_{u}_{b}_%=")
    masked(e, [f"global_store_dwordx4 v{V_STOFF}, {quad(V_SR)}, s[{S_OUT}:{S_OUT + 1}]",
               f"global_store_dwordx4 v{V_STOFF}, {quad(V_SR + 4)}, s[{S_OUT}:{S_OUT + 1}] offset:16"])
    bump(e, S_OUT, 32)
    e(f"s_mov_b32 s{S_NCAP}, 0")
    e(f"CAPK_{u}_{b}_%=:")
    e(f"s_add_i32 s{S_NEXT}, s{S_NEXT}, s{S_STEP}")
    e(f"s_cmp_lt_i32 s{S_NEXT}, 0")
    e(f"s_cbranch_scc1 CAP_{u}_{b}_%=")
    e(f"s_branch CAPRET_{u}_{b}_%=")


def gen_bits(g_, b):
    """bits variant, batch b (windows 8b .. 8b+7 of the tile, accumulators in V_ACC): the compare (cutoff <= score; false
    for a NaN) leaves the window's bit in VCC, and an add-with-carry shifts it into the tile's dword,
    bits = 2 * bits + carry: two instructions per window.  The first window comes out on top: gen_tile reverses the
    dword (v_bfrev_b32) when the tile is complete."""
    e = g_.emit
    for j in range(8):
        e(f"v_cmp_le_f64_e32 vcc, s[{S_CUT}:{S_CUT + 1}], {pair(V_ACC + 2 * j)}")
        src = V_ZERO if (b == 0 and j == 0) else V_BITS      # the tile's first window starts the dword
        e(f"v_addc_co_u32_e32 v{V_BITS}, vcc, v{src}, v{src}, vcc")


def term_loads(e, tb, tile_off):
    """this lane's two terms of each of its two windows (passes) of a tile's pair table, `tile_off` bytes from the pointers"""
    if "nocomb" in ABL:
        return
    for ps in range(2):
        e(f"global_load_dwordx2 {pair(tb + 4 * ps)}, v{V_CT}, s[{S_PTT}:{S_PTT + 1}] offset:{tile_off + 128 * ps}")
        e(f"global_load_dwordx2 {pair(tb + 4 * ps + 2)}, v{V_CL}, s[{S_PTL}:{S_PTL + 1}] offset:{tile_off + 128 * ps}")


def comb_writes(g_, tb, slot):
    if "nocomb" in ABL:
        return
    for ps in range(2):
        g_.lds(f"ds_write_b128 v{V_CW}, {quad(tb + 4 * ps)} offset:{COMB_BASE + slot * COMB_SLOT + 1024 * ps}")


# vector-memory operations a tile issues in program order (loads return in order; a store between them only makes a
# counted wait more conservative): 4 term loads, then 4 genotype words
N_TERM, N_WORDS = 4, 4


def gen_tile(g_, u):
    """tile t = UNROLL i + u of the loop"""
    e = g_.emit
    lo_next = "B" if u % 2 == 0 else "A"
    nwords = 0 if "nowords" in ABL else N_WORDS
    if "nobar" not in ABL:
        e("s_barrier")
    # ---- requests: this lane's terms of the pair table of tile t + 3; the genotype words 7 and 8 of the tile (tile
    #      t + 2 funnel-shifts word 7 for its look-ahead)
    term_loads(e, V_TB[u % 2], (u % 4) * 1024)
    if u % 4 == 3:
        bump(e, S_PTL, 4096)
        bump(e, S_PTT, 4096)
    e(f"v_alignbit_b32 v{V_FUN}, v{word(V_WL, 2 * u + 2)}, v{word(V_WL, 2 * u + 1)}, s{S_SHL}")
    e(f"v_alignbit_b32 v{V_FUN + 1}, v{word(V_WT, 2 * u + 2)}, v{word(V_WT, 2 * u + 1)}, s{S_SHT}")
    if nwords:
        for ring, ptr in ((V_WL, S_PLW), (V_WT, S_PTW)):
            for i in (7, 8):
                e(f"global_load_dword v{word(ring, 2 * u + i)}, v{V_LANE4}, s[{ptr}:{ptr + 1}] offset:{(2 * u + i - 8) * 256}")
    if u == UNROLL - 1:
        bump(e, S_PLW, 2 * UNROLL * 256)
        bump(e, S_PTW, 2 * UNROLL * 256)
    # ---- the tile's second word (windows 16..31) of both streams: code words
    gen_codes(e, "H")

    # ---- 8 groups; on entry the look-ups of group 0 are in flight (issued beside group 7 of the previous tile)
    g_.issued = 4
    g_.complete = 0
    r_prev = 4                                   # id of the last look-up of R(0)
    for g in range(8):
        if g == 2:
            # the pair table of tile t + 2: its terms were requested at the start of tile t - 1; everything issued
            # since may stay in flight (the words of tile t - 1, this tile's requests)
            if "nocomb" not in ABL:
                e(f"s_waitcnt vmcnt({nwords + N_TERM + nwords})")
            comb_writes(g_, V_TB[(u + 1) % 2], (u + 2) % NSLOT)
        if g == 6:
            # first word of the next tile: funnel shift + code words (its genotype words were requested two and three
            # tiles ago; the counted wait of group 2 covers them)
            e(f"v_alignbit_b32 v{V_FUN}, v{word(V_WL, 2 * u + 3)}, v{word(V_WL, 2 * u + 2)}, s{S_SHL}")
            e(f"v_alignbit_b32 v{V_FUN + 1}, v{word(V_WT, 2 * u + 3)}, v{word(V_WT, 2 * u + 2)}, s{S_SHT}")
            gen_codes(e, lo_next)
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
    if g_.bits:
        # the tile's 32 bits: a dword per lane, kept until the iteration's eight are complete: 32 aligned bytes per lane
        e(f"v_bfrev_b32_e32 v{V_SR + u}, v{V_BITS}")
        if u == UNROLL - 1:
            masked(e, [f"global_store_dwordx4 v{V_STOFF}, {quad(V_SR)}, s[{S_OUT}:{S_OUT + 1}]",
                       f"global_store_dwordx4 v{V_STOFF}, {quad(V_SR + 4)}, s[{S_OUT}:{S_OUT + 1}] offset:16"])
            bump(e, S_OUT, 32)


def gen_idle_tile(g_, u):
    """a wave without a block: the barrier and its quarter of the pair tables"""
    e = g_.emit
    if "nobar" not in ABL:
        e("s_barrier")
    term_loads(e, V_TB[u % 2], (u % 4) * 1024)
    if u % 4 == 3:
        bump(e, S_PTL, 4096)
        bump(e, S_PTT, 4096)
    if "nocomb" not in ABL:
        e(f"s_waitcnt vmcnt({N_TERM})")
    comb_writes(g_, V_TB[(u + 1) % 2], (u + 2) % NSLOT)
    e("s_waitcnt lgkmcnt(0)")


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
    e(f"v_lshlrev_b32_e32 v{V_LANE4}, 2, %[lane]")
    # this lane's place in a tile's pair table: window 8 wave + lane / 16 (second pass: + 4), pair lane % 16 = 4 g_out + g_in
    e(f"s_lshl_b32 s{S_TMP}, %[wave], 3")
    e(f"v_lshrrev_b32_e32 v{V_CW}, 4, %[lane]")
    e(f"v_add_u32_e32 v{V_CW}, s{S_TMP}, v{V_CW}")                    # the window
    e(f"v_lshlrev_b32_e32 v{V_CT}, 5, v{V_CW}")                        # its term row, 32 B
    e(f"v_and_b32_e32 v{V_CL}, 3, %[lane]")
    e(f"v_lshl_add_u32 v{V_CL}, v{V_CL}, 3, v{V_CT}")                  # + 8 g_in
    e(f"v_bfe_u32 v{V_S}, %[lane], 2, 2")
    e(f"v_lshl_add_u32 v{V_CT}, v{V_S}, 3, v{V_CT}")                   # + 8 g_out
    e(f"v_and_b32_e32 v{V_S}, 15, %[lane]")
    e(f"v_lshlrev_b32_e32 v{V_CW}, 8, v{V_CW}")
    e(f"v_lshl_add_u32 v{V_CW}, v{V_S}, 4, v{V_CW}")                   # 256 window + 16 pair
    # the pair tables of tiles 0 and 1, the terms of tile 2 on their way
    term_loads(e, V_TB[0], 0)
    term_loads(e, V_TB[1], 1024)
    if "nocomb" not in ABL:
        e("s_waitcnt vmcnt(0)")
    comb_writes(g_, V_TB[0], 0)
    comb_writes(g_, V_TB[1], 1)
    e("s_waitcnt lgkmcnt(0)")                    # (the writes have taken their registers)
    term_loads(e, V_TB[1], 2048)
    bump(e, S_PTL, TERM_AHEAD * 1024)            # the loop's term offsets are (u % 4) * 1024 from here
    bump(e, S_PTT, TERM_AHEAD * 1024)
    e("s_cmp_eq_u32 %[active], 0")
    e("s_cbranch_scc1 IDLE_%=")
    e(f"s_mov_b64 s[{S_PLW}:{S_PLW + 1}], %[plw]")
    e(f"s_mov_b64 s[{S_PTW}:{S_PTW + 1}], %[ptw]")
    e(f"s_mov_b64 s[{S_OUT}:{S_OUT + 1}], %[out]")
    e(f"s_mov_b32 s{S_NEXT}, %[next]")
    e(f"s_mov_b32 s{S_STEP}, %[step]")
    e(f"s_mov_b32 s{S_SHL}, %[shl]")
    e(f"s_mov_b32 s{S_SHT}, %[sht]")
    e(f"s_movk_i32 s{S_F0}, 0xf0")
    e(f"s_mov_b32 s{S_MCC}, 0xcccccccc")
    e(f"s_mov_b32 s{S_NCAP}, 0")
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
    bump(e, S_PLW, 2048)       # the loop's word offsets are relative to word 8 of the iteration's first tile
    bump(e, S_PTW, 2048)
    e("s_waitcnt vmcnt(0)")
    if "nobar" not in ABL:
        e("s_barrier")         # the pair tables of tiles 0 and 1 are complete
    # first word of tile 0 -> LO_A, and its group 0 look-ups
    e(f"v_alignbit_b32 v{V_FUN}, v{word(V_WL, 1)}, v{word(V_WL, 0)}, s{S_SHL}")
    e(f"v_alignbit_b32 v{V_FUN + 1}, v{word(V_WT, 1)}, v{word(V_WT, 0)}, s{S_SHT}")
    gen_codes(e, "A")
    # group 0 of tile 0 = "group 8" of a tile u = 3 (next LO = A, next slot = 0)
    gen_E(g_, 3, 8)
    gen_R(g_, 3, 8, (0, 1, 2, 3))
    # (the terms of tile 2 were requested before this wave's 14 words: the loop's first counted wait sees them as tile -1's)
    e("LOOP_%=:")
    for u in range(UNROLL):
        gen_tile(g_, u)
    e(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    e(f"s_cmp_lg_u32 s{S_CNT}, 0")
    e("s_cbranch_scc1 LOOP_%=")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    if not bits and "nocapture" not in ABL:
        # the samples still kept (fewer than four): 8 B each
        for k in range(3):
            e(f"s_cmp_le_u32 s{S_NCAP}, {k}")
            e("s_cbranch_scc1 FLUSHED_%=")
            masked(e, [f"global_store_dwordx2 v{V_STOFF}, {pair(V_SR + 2 * k)}, s[{S_OUT}:{S_OUT + 1}] offset:{8 * k}"])
        e("FLUSHED_%=:")
        e(f"s_lshl_b32 s{S_TMP}, s{S_NCAP}, 3")
        e(f"s_add_u32 s{S_OUT}, s{S_OUT}, s{S_TMP}")
        e(f"s_addc_u32 s{S_OUT + 1}, s{S_OUT + 1}, 0")
    e(f"v_mov_b64 %[acc], {pair(V_ACC + 14)}")
    e(f"s_mov_b32 %[next_out], s{S_NEXT}")
    e(f"s_mov_b64 %[out_out], s[{S_OUT}:{S_OUT + 1}]")
    e("s_branch DONE_%=")
    if "nocapture" not in ABL and not bits:
        for u in range(UNROLL):
            for b in range(4):
                gen_capture(g_, u, b)
    e("IDLE_%=:")
    if "nobar" not in ABL:
        e("s_barrier")
    e("IDLE_LOOP_%=:")
    for u in range(UNROLL):
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
        f.write(f"#define GARLIC_FEED_UNROLL {UNROLL}\n")
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
