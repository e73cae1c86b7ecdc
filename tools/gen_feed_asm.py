#!/usr/bin/env python3
"""Generates garlic_amd/csrc/feed_loop_gfx950.inc: the hand-scheduled interior loop of lod_feed_kernel /
lod_bits_kernel (feed_kernel.hpp) -- the window recurrence of src/garlic-roh.cpp:92-100 with every wavefront a
chain of its own (gfx950, wave64).

Why by hand: the length of the longest run of windows x the pace of ONE wave is the kernel's critical path (10M SNPs:
440k windows in one run), and what a wave's pace is made of had to be measured (tools/exp/r4_variants.sh):

  * a wavefront alone issues one instruction per ~4 cycles whatever the instruction is (tools/ubench/issue_rates.hip);
  * a VECTOR-MEMORY instruction costs the CU ~50 cycles of issue whatever it moves (4 B or 16 B per lane): round 3's loop
    (6 per tile and wave) and this round's first form (10: the lanes' own genotype words one row at a time, the term rows
    gathered 8 B per lane) spent 60 % of their time there, alone on a CU and sixteen to a CU alike.  So everything comes in
    1-KB pieces -- one global_load_dwordx4 per 32 term rows or 4 genotype word rows -- through LDS: 1.5 vector-memory
    instructions per tile and wave;
  * look-ups: per window ONE ds_read_b128 {t_out[g_out], t_in[g_in]} from the window's 16-entry pair table, its offset
    16 * (4 g_out + g_in) from ONE SDWA instruction (byte select + AND / shift into byte 0 of a word of 4-bit codes; the
    codes of 16 SNPs cost 4 instructions: v_lshl + v_bfi of the two funnel-shifted genotype words, even / odd SNPs).

  per window and lane (lane = individual):  SDWA extract | ds_read_b128 | v_add_f64 x 2: acc = (acc - t_out) + t_in, two
  roundings as src/garlic-roh.cpp:98-100 | bits variant: v_cmp_le_f64 + v_addc_co_u32 (the window's coverage bit)

  tile = 32 windows; the workgroup's four waves (four 64-individual blocks of one run) share, in LDS:
    raw rows     {lod(0), lod(1), lod(2), +0.0} of the 32 entering and the 32 leaving SNPs of a tile, 2 x 1 KB, two slots:
                 tile T's rows are requested by wave T % 4 during tile T - 5 (two global_load_dwordx4), written during
                 tile T - 3;
    pair tables  32 windows x 16 pairs x 16 B = 8 KB per tile, four slots: tile T's table is built during tile T - 2, a
                 quarter per wave (4 ds_read_b64 from the raw rows, 2 ds_write_b128), read during tile T and, by the
                 look-ahead, at the end of tile T - 1;
    one s_barrier per tile is the protocol of both rings;
  and each wave, for itself: its lanes' packed genotype words, 4 word rows (1 KB) per global_load_dwordx4, entering stream
  in odd tiles, leaving stream in even ones, turned lane-wise through 1 KB of LDS (ds_write_b128, 2 ds_read2st64_b32)
  into 8-word register rings two tiles after they were requested.

  samples (feed)   a sampled locus (every step-th, src/garlic-data.cpp:2036) is picked from the batch's 8 accumulators
                   with s_set_gpr_idx and kept; eight of them leave as one 64-byte piece per lane;
  bits             a tile's 32 bits are a dword per lane; eight tiles' dwords leave as one aligned 32-byte piece per lane.

The block runs 8 n tiles, all of them interior (every window a rolling update); lod_feed_kernel runs the other tiles
through its compiler-generated path, which needs no ring: it reads the term table itself.  A wave without a block
(`active` = 0) takes its turns as loader and builds its quarter of the tables.

Environment hooks for experiments: GARLIC_FEED_ABLATE (nodp, nolds, nobar, nocapture, nowords, nocomb), GARLIC_FEED_LOOK,
GARLIC_FEED_WAITEVERY.
"""
import os

ABL = os.environ.get("GARLIC_FEED_ABLATE", "")
LOOK = int(os.environ.get("GARLIC_FEED_LOOK", "6"))           # a window's look-up is issued this many windows ahead
WAITEVERY = int(os.environ.get("GARLIC_FEED_WAITEVERY", "4"))  # windows per counted s_waitcnt lgkmcnt

# ---- LDS map (bytes, workgroup-relative; feed_kernel.hpp takes GARLIC_FEED_LDS_TOTAL from the generated file)
LDS_MISC = 0            # item word etc. (compiler-generated code only)
COMB_BASE = int(os.environ.get("GARLIC_FEED_COMB_BASE", "1024"))   # 0: the item word shares slot 0's first bytes (only written between items)
COMB_SLOT = 8192        # 32 windows x 16 pairs x 16 B
NSLOT = 4
RAW_BASE = COMB_BASE + NSLOT * COMB_SLOT
RAW_SLOT = 2048         # entering SNPs' rows at + 0, leaving SNPs' rows at + 1024
WST_BASE = RAW_BASE + 2 * RAW_SLOT      # a wave's 4 genotype word rows on their way into its registers: 1 KB per wave
LDS_TOTAL = WST_BASE + 4 * 1024
UNROLL = 8              # tiles per loop iteration
RAW_AHEAD = 5           # the raw rows of tile t + 5 are requested during tile t

# ---- fixed VGPRs (clobbered by the block)
V_WL = 20               # lead word ring, 8 registers: word row i of the stream (from the loop's first tile) at i % 8
V_WT = 28               # trail word ring
V_H = 36                # 4-bit codes 4 g_out + g_in: LO_A {even, odd} | LO_B {even, odd} | HI {even, odd}
V_FUN = 42              # funnel-shifted genotype words (2 registers: entering, leaving)
V_BUF = 44              # look-up buffers of NBUF windows in flight or in use: {t_out, t_in} (4 registers each); a window's offset is extracted into its first register
NBUF = int(os.environ.get("GARLIC_FEED_NBUF", "8"))     # (a power of two: a tile's window 32 takes the buffer of the next tile's window 0)
assert NBUF in (4, 8)
_SH = 4 * (8 - NBUF)    # registers the map below moves down by
V_ACC = 76 - _SH        # 8 accumulators of the current batch (16 registers)
V_TB = 92 - _SH         # builder: 2 passes x {t_out, t_in} between the raw rows and the pair table (8 registers)
V_TL = 100 - _SH        # this wave's eighth of a tile's raw rows on its way, 8 B per lane: two requests in flight (2 x 2 registers)
V_L8 = 104 - _SH        # lane * 8
V_RAWW = 105 - _SH      # ... its place in a raw slot: RAW_BASE + (wave / 2) * 1024 + (wave % 2) * 512 + lane * 8
V_WSL, V_WST = 108 - _SH, 112 - _SH   # 4 word rows of the entering / leaving stream on their way (4 registers each)
V_L16 = 116 - _SH       # lane * 16
V_L4 = 117 - _SH        # lane * 4
V_CT = 118 - _SH        # this lane's leaving-SNP term in a tile's raw rows:  32 * (8 wave + lane / 16) + 8 * ((lane % 16) / 4)
V_CL = 119 - _SH        # this lane's entering-SNP term:                      32 * (8 wave + lane / 16) + 8 * (lane % 4)
V_CW = 120 - _SH        # its pair in the table:                              256 * (8 wave + lane / 16) + 16 * (lane % 16)
V_STOFF = 121 - _SH     # the lane's row * row pitch (bytes) of the sample / bit matrix
V_S = 122 - _SH         # the sampled accumulator (2 registers)
V_SR = 124 - _SH        # feed: KEEP = 8 kept samples (16 registers); bits: the dwords of the iteration's 8 tiles
KEEP = 8
V_WSA = 140 - _SH       # this wave's word staging area, the lane's column: WST_BASE + wave * 1024 + lane * 4
V_WSW = 141 - _SH       # ... the lane's 16 bytes of it:                      WST_BASE + wave * 1024 + lane * 16
CLOBBER_V = list(range(20, 142 - _SH))
# ---- fixed SGPRs
S_PLW, S_PTW = 40, 42   # genotype word streams: address of the block's word row 0 of the loop's first tile, + the loop's bias
S_PTR = 44              # this wave's eighth of the term rows (waves 0, 1: entering SNPs, 2, 3: leaving; 16 rows each), RAW_AHEAD tiles ahead of the iteration's first tile
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
S_NCAP = 65             # feed: samples kept (0 .. KEEP - 1)
S_WAVE = 66
CLOBBER_S = list(range(40, 67))
# bits variant (GARLIC_FEED_BITS_LOOP_ASM): instead of sampled scores the loop leaves ONE BIT per window and lane --
# score >= cutoff -- 32 of them per tile in V_BITS; V_ZERO holds zero, the window's bit goes from VCC into the dword
# by an add-with-carry
V_BITS, V_ZERO = V_S, V_S + 1


class Gen:
    """instruction list + a model of the in-order LGKM counter (LDS operations every wave issues; what only some waves
    issue -- the loader's writes -- is left out: an operation the model does not know makes a counted wait wait longer,
    never shorter)"""

    def __init__(self, bits=False):
        self.out = []
        self.issued = 0
        self.complete = 0
        self.bits = bits
        self.last_sdwa_dst = None
        self.labels = 0

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
        assert 0 <= n <= 15
        self.emit(f"s_waitcnt lgkmcnt({n})")
        self.complete = op

    def label(self, stem):
        self.labels += 1
        return f"{stem}_{self.labels}_%="


def pair(r):
    assert r % 2 == 0
    return f"v[{r}:{r + 1}]"


def quad(r):
    assert r % 2 == 0
    return f"v[{r}:{r + 3}]"


def word(ring, i):
    return ring + i % 8


def hreg(which, odd):
    """which: 'A' / 'B' (the LO words of even / odd tiles) or 'H'"""
    return V_H + {"A": 0, "B": 2, "H": 4}[which] + odd


def bump(e, ptr, n):
    e(f"s_add_u32 s{ptr}, s{ptr}, {n}")
    e(f"s_addc_u32 s{ptr + 1}, s{ptr + 1}, 0")


def gen_codes(e, which):
    """the two code words of 16 SNPs from the funnel-shifted words V_FUN (entering) and V_FUN + 1 (leaving):
    nibble i of the even word = 4 g_out + g_in of SNP 2 i, of the odd word of SNP 2 i + 1"""
    he, ho = hreg(which, 0), hreg(which, 1)
    e(f"v_lshlrev_b32_e32 v{he}, 2, v{V_FUN + 1}")
    e(f"v_lshrrev_b32_e32 v{ho}, 2, v{V_FUN}")
    e(f"v_bfi_b32 v{he}, s{S_MCC}, v{he}, v{V_FUN}")          # (mask & out << 2) | (~mask & in)
    e(f"v_bfi_b32 v{ho}, s{S_MCC}, v{V_FUN + 1}, v{ho}")      # (mask & out) | (~mask & in >> 2)


def win_source(u, x):
    """window x of unrolled tile u (x >= 32: window x - 32 of the next tile): its code words, the SNP's place in them, the
    LDS slot of its pair table and its index in that tile"""
    lo_this = "A" if u % 2 == 0 else "B"
    lo_next = "B" if u % 2 == 0 else "A"
    if x >= 32:
        assert x - 32 < 16
        return lo_next, x - 32, (u + 1) % NSLOT, x - 32
    return (lo_this if x < 16 else "H"), x % 16, u % NSLOT, x


def buf_of(x):
    """look-up buffer of window x: {t_out, t_in}, four registers; the offset is extracted into the first"""
    return V_BUF + 4 * (x % NBUF)


def gen_E(g_, u, x):
    """the look-up offset 16 * (4 g_out + g_in) of window x: SNP n of its word pair is nibble n / 2 of the even (n even) /
    odd word = byte n / 4, low (n % 4 < 2) / high nibble"""
    which, n, _, _ = win_source(u, x)
    src = hreg(which, n & 1)
    dst = buf_of(x)
    b = n >> 2
    if n & 2:
        g_.sdwa(f"v_and_b32_sdwa v{dst}, s{S_F0}, v{src} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_{b}", dst)
    else:
        g_.sdwa(f"v_lshlrev_b32_sdwa v{dst}, 4, v{src} dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_{b}", dst)


def gen_R(g_, u, x):
    """the look-up of window x; returns its id in the LGKM model"""
    _, _, slot, j = win_source(u, x)
    r = buf_of(x)
    if "nolds" in ABL:
        g_.emit("s_nop 0")
        return g_.issued
    return g_.lds(f"ds_read_b128 {quad(r)}, v{r} offset:{COMB_BASE + slot * COMB_SLOT + 256 * j}", r)


def gen_add(g_, x, half):
    """half 0: acc(window x) = acc(window x - 1) - t_out; half 1: += t_in"""
    b = buf_of(x)
    dst = V_ACC + 2 * (x % 8)
    prev = V_ACC + 2 * ((x - 1) % 8)
    if "nodp" in ABL:      # (timing experiment: the same adds without the chain through the windows)
        prev = b + 2
    if half == 0:
        g_.emit(f"v_add_f64 {pair(dst)}, {pair(prev)}, -{pair(b)}")
    else:
        g_.emit(f"v_add_f64 {pair(dst)}, {pair(dst)}, {pair(b + 2)}")


def masked(e, lines):
    e(f"s_mov_b64 exec, s[{S_MASK}:{S_MASK + 1}]")          # individuals without a row: not stored
    for ln in lines:
        e(ln)
    e(f"s_mov_b64 exec, s[{S_EXEC}:{S_EXEC + 1}]")


def gen_capture(g_, u, b):
    """out of line: the sampled accumulator(s) of batch b of tile u.  S_NEXT has gone below zero (mod 2^32).  The
    sample joins the kept ones; KEEP = 8 leave together as 64 bytes per lane."""
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
    e(f"s_cmp_lg_u32 s{S_NCAP}, {KEEP}")
    e(f"s_cbranch_scc1 CAPK_{u}_{b}_%=")
    masked(e, [f"global_store_dwordx4 v{V_STOFF}, {quad(V_SR + 4 * k)}, s[{S_OUT}:{S_OUT + 1}] offset:{16 * k}" for k in range(KEEP // 2)])
    bump(e, S_OUT, 8 * KEEP)
    e(f"s_mov_b32 s{S_NCAP}, 0")
    e(f"CAPK_{u}_{b}_%=:")
    e(f"s_add_i32 s{S_NEXT}, s{S_NEXT}, s{S_STEP}")
    e(f"s_cmp_lt_i32 s{S_NEXT}, 0")
    e(f"s_cbranch_scc1 CAP_{u}_{b}_%=")
    e(f"s_branch CAPRET_{u}_{b}_%=")


def capture_check(e, u, b):
    e(f"s_sub_u32 s{S_NEXT}, s{S_NEXT}, 8")          # borrow: a sampled locus among the batch's 8 windows
    e(f"s_cbranch_scc1 CAP_{u}_{b}_%=")
    e(f"CAPRET_{u}_{b}_%=:")


def bit_compare(e, j):
    """window j of the tile: cutoff <= score (false for a NaN) into VCC"""
    e(f"v_cmp_le_f64_e32 vcc, s[{S_CUT}:{S_CUT + 1}], {pair(V_ACC + 2 * (j % 8))}")


def bit_into_dword(e, first):
    """bits = 2 * bits + VCC (the tile's first window starts the dword; it comes out on top: the dword is reversed
    when the tile is complete)"""
    src = V_ZERO if first else V_BITS
    e(f"v_addc_co_u32_e32 v{V_BITS}, vcc, v{src}, v{src}, vcc")


# ------------------------------------------------------------------------------------------------ raw rows, pair tables
def raw_request(g_, k, off):
    """every wave requests its eighth of a tile's raw rows (half of one stream's 32 rows: 8 B per lane), `off` bytes from its
    term pointer, into staging pair k"""
    if "nocomb" in ABL:
        return
    g_.emit(f"global_load_dwordx2 {pair(V_TL + 2 * k)}, v{V_L8}, s[{S_PTR}:{S_PTR + 1}] offset:{off}")


def raw_write(g_, k, slot):
    """... and writes it into raw slot `slot`"""
    if "nocomb" in ABL:
        return
    g_.lds(f"ds_write_b64 v{V_RAWW}, {pair(V_TL + 2 * k)} offset:{slot * RAW_SLOT}")


def build_reads(g_, raw_slot):
    """this lane's two terms of each of its two windows (passes) of a tile's pair table from the raw rows"""
    if "nocomb" in ABL:
        return g_.issued
    base = RAW_BASE + raw_slot * RAW_SLOT
    last = 0
    for ps in range(2):
        g_.lds(f"ds_read_b64 {pair(V_TB + 4 * ps)}, v{V_CT} offset:{base + 1024 + 128 * ps}")
        last = g_.lds(f"ds_read_b64 {pair(V_TB + 4 * ps + 2)}, v{V_CL} offset:{base + 128 * ps}")
    return last


def build_writes(g_, comb_slot):
    if "nocomb" in ABL:
        return
    for ps in range(2):
        g_.lds(f"ds_write_b128 v{V_CW}, {quad(V_TB + 4 * ps)} offset:{COMB_BASE + comb_slot * COMB_SLOT + 1024 * ps}")


# ------------------------------------------------------------------------------------------------ genotype words
def words_stage(g_, u):
    """tile start, the stream whose turn it is (odd tiles: entering, even: leaving): the 4 word rows requested two tiles ago
    go into this wave's staging KB.  (words_into_ring moves them on into the ring: the rows they replace were last used by
    the previous tile, or by this tile's funnel shifts, which the caller issues in between.)"""
    if "nowords" in ABL:
        return
    ws = V_WSL if u % 2 == 1 else V_WST
    g_.lds(f"ds_write_b128 v{V_WSW}, {quad(ws)}")


def words_request(g_, u):
    """... and the next 4 rows are requested"""
    if "nowords" in ABL:
        return
    ws, ptr = (V_WSL, S_PLW) if u % 2 == 1 else (V_WST, S_PTW)
    g_.emit(f"global_load_dwordx4 {quad(ws)}, v{V_L16}, s[{ptr}:{ptr + 1}] offset:{(u // 2) * 1024}")


def words_into_ring(g_, u):
    if "nowords" in ABL:
        return
    lead = u % 2 == 1
    batch = (u + 3) // 2 if lead else (u + 2) // 2        # (mod 2: which half of the ring)
    ring = V_WL if lead else V_WT
    r0 = ring + 4 * (batch % 2)
    g_.lds(f"ds_read2st64_b32 {pair(r0)}, v{V_WSA} offset1:1")
    g_.lds(f"ds_read2st64_b32 {pair(r0 + 2)}, v{V_WSA} offset0:2 offset1:3")


# ------------------------------------------------------------------------------------------------ a tile
def gen_tile(g_, u):
    """tile t = UNROLL i + u of the loop"""
    e = g_.emit
    lo_next = "B" if u % 2 == 0 else "A"
    if "nobar" not in ABL:
        e("s_barrier")
    g_.issued = LOOK                 # on entry the look-ups of windows 0 .. LOOK - 1 are in flight (issued by the previous tile)
    g_.complete = 0
    rid = {x: x + 1 for x in range(LOOK)}
    # what this wave requested two tiles ago -- 4 genotype word rows of the stream whose turn it is, its eighth of the raw
    # rows of tile t + 3 -- goes into LDS; the requests of the previous tile stay in flight (the counter is in order and
    # every wave issues the same requests: the count is exact); then this tile's requests: word rows, raw rows of tile t + 5
    nreq = (0 if "nowords" in ABL else 1) + (0 if "nocomb" in ABL else 1)
    e(f"s_waitcnt vmcnt({nreq})")
    words_stage(g_, u)
    raw_write(g_, u % 2, (u + 3) % 2)
    words_request(g_, u)
    raw_request(g_, u % 2, (u % 4) * 1024)
    if u % 4 == 3:
        bump(e, S_PTR, 4096)
    if u == UNROLL - 1:
        bump(e, S_PLW, 4096)
        bump(e, S_PTW, 4096)
    # ---- the tile's second word (windows 16..31) of both streams: funnel shifts, code words; then the ring may change
    e(f"v_alignbit_b32 v{V_FUN}, v{word(V_WL, 2 * u + 2)}, v{word(V_WL, 2 * u + 1)}, s{S_SHL}")
    e(f"v_alignbit_b32 v{V_FUN + 1}, v{word(V_WT, 2 * u + 2)}, v{word(V_WT, 2 * u + 1)}, s{S_SHT}")
    words_into_ring(g_, u)
    gen_codes(e, "H")

    # ---- 32 windows, one at a time, every dependent pair of the chain separated by independent work:
    #   offset of window w + LOOK | acc - t_out | feed: its look-up; bits: the previous window's bit into the dword | + t_in |
    #   bits: the look-up, the compare
    # A look-up is waited for LOOK - WAITEVERY + 1 windows after it was issued at the least, WAITEVERY windows per s_waitcnt.
    rid_raw = None
    for w in range(32):
        if w == 8:
            rid_raw = build_reads(g_, u % 2)                 # pair table of tile t + 2 from raw slot (t + 2) % 2
        if w == 12:
            g_.wait_lds(rid_raw)
            build_writes(g_, (u + 2) % NSLOT)
        if w == 24:
            # first word of the next tile: funnel shift + code words
            e(f"v_alignbit_b32 v{V_FUN}, v{word(V_WL, 2 * u + 3)}, v{word(V_WL, 2 * u + 2)}, s{S_SHL}")
            e(f"v_alignbit_b32 v{V_FUN + 1}, v{word(V_WT, 2 * u + 3)}, v{word(V_WT, 2 * u + 2)}, s{S_SHT}")
            gen_codes(e, lo_next)
        if w % WAITEVERY == 0:
            g_.wait_lds(rid[w + WAITEVERY - 1])
        gen_E(g_, u, w + LOOK)
        gen_add(g_, w, 0)
        if g_.bits:
            if w > 0:
                bit_into_dword(e, first=(w == 1))
            gen_add(g_, w, 1)
            rid[w + LOOK] = gen_R(g_, u, w + LOOK)
            bit_compare(e, w)
        else:
            rid[w + LOOK] = gen_R(g_, u, w + LOOK)
            gen_add(g_, w, 1)
        if w % 8 == 7 and not g_.bits and "nocapture" not in ABL:
            capture_check(e, u, w // 8)
    if g_.bits:
        bit_into_dword(e, first=False)           # window 31
        # the tile's 32 bits: a dword per lane, kept until the iteration's eight are complete: 32 aligned bytes per lane
        e(f"v_bfrev_b32_e32 v{V_SR + u}, v{V_BITS}")
        if u == UNROLL - 1:
            masked(e, [f"global_store_dwordx4 v{V_STOFF}, {quad(V_SR)}, s[{S_OUT}:{S_OUT + 1}]",
                       f"global_store_dwordx4 v{V_STOFF}, {quad(V_SR + 4)}, s[{S_OUT}:{S_OUT + 1}] offset:16"])
            bump(e, S_OUT, 32)


def gen_idle_tile(g_, u):
    """a wave without a block: the barrier, its eighth of the raw rows, its quarter of the pair tables"""
    e = g_.emit
    if "nobar" not in ABL:
        e("s_barrier")
    g_.issued = 0
    g_.complete = 0
    if "nocomb" not in ABL:
        e("s_waitcnt vmcnt(1)")
    raw_write(g_, u % 2, (u + 3) % 2)
    raw_request(g_, u % 2, (u % 4) * 1024)
    if u % 4 == 3:
        bump(e, S_PTR, 4096)
    g_.wait_lds(build_reads(g_, u % 2))
    build_writes(g_, (u + 2) % NSLOT)
    e("s_waitcnt lgkmcnt(0)")


def gen_all(bits=False):
    g_ = Gen(bits)
    e = g_.emit
    assert 1 <= LOOK < NBUF and 32 % WAITEVERY == 0 and WAITEVERY <= LOOK
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e(f"s_mov_b32 s{S_CNT}, %[niter]")
    e(f"s_mov_b32 s{S_WAVE}, %[wave]")
    # this wave's eighth of a tile's raw rows: waves 0, 1 the entering SNPs' rows 0-15 / 16-31, waves 2, 3 the leaving SNPs'
    e(f"s_cmp_lt_u32 s{S_WAVE}, 2")
    e(f"s_cselect_b64 s[{S_PTR}:{S_PTR + 1}], %[ptl], %[ptt]")
    e(f"s_and_b32 s{S_TMP}, s{S_WAVE}, 1")
    e(f"s_lshl_b32 s{S_TMP}, s{S_TMP}, 9")
    e(f"s_add_u32 s{S_PTR}, s{S_PTR}, s{S_TMP}")
    e(f"s_addc_u32 s{S_PTR + 1}, s{S_PTR + 1}, 0")
    e(f"s_lshr_b32 s{S_IDX}, s{S_WAVE}, 1")
    e(f"s_lshl_b32 s{S_IDX}, s{S_IDX}, 10")
    e(f"s_add_u32 s{S_TMP}, s{S_TMP}, s{S_IDX}")                       # (wave / 2) * 1024 + (wave % 2) * 512
    e(f"s_add_u32 s{S_TMP}, s{S_TMP}, {RAW_BASE}")
    e(f"v_lshlrev_b32_e32 v{V_L8}, 3, %[lane]")
    e(f"v_add_u32_e32 v{V_RAWW}, s{S_TMP}, v{V_L8}")
    e(f"v_lshlrev_b32_e32 v{V_L4}, 2, %[lane]")
    e(f"v_lshlrev_b32_e32 v{V_L16}, 4, %[lane]")
    # this lane's place in a tile's pair table: window 8 wave + lane / 16 (second pass: + 4), pair lane % 16 = 4 g_out + g_in
    e(f"s_lshl_b32 s{S_TMP}, s{S_WAVE}, 3")
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
    e(f"s_lshl_b32 s{S_TMP}, s{S_WAVE}, 10")
    e(f"s_add_u32 s{S_TMP}, s{S_TMP}, {WST_BASE}")
    e(f"v_add_u32_e32 v{V_WSA}, s{S_TMP}, v{V_L4}")
    e(f"v_add_u32_e32 v{V_WSW}, s{S_TMP}, v{V_L16}")
    # ---- the pair tables of tiles 0 and 1
    raw_request(g_, 0, 0)
    raw_request(g_, 1, 1024)
    if "nocomb" not in ABL:
        e("s_waitcnt vmcnt(0)")
    raw_write(g_, 0, 0)
    raw_write(g_, 1, 1)
    e("s_waitcnt lgkmcnt(0)")
    if "nobar" not in ABL:
        e("s_barrier")
    for t in range(2):
        g_.wait_lds(build_reads(g_, t))
        build_writes(g_, t)
    e("s_waitcnt lgkmcnt(0)")
    if "nobar" not in ABL:
        e("s_barrier")               # both tables complete; the raw slots are free
    # ---- the raw rows of tile 2 into slot 0
    raw_request(g_, 0, 2048)
    if "nocomb" not in ABL:
        e("s_waitcnt vmcnt(0)")
    raw_write(g_, 0, 0)
    e("s_waitcnt lgkmcnt(0)")
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
    # word rows 0..7 of the entering and 0..3 of the leaving stream row by row (tile t funnel-shifts rows 2t .. 2t + 3);
    # the loop brings 4 rows at a time: the leaving stream's rows 4..7 (into the ring in tile 0) and the entering
    # stream's rows 8..11 (tile 1) are requested here -- and the raw rows of tiles 3 and 4 (written by the loop's tiles 0
    # and 1), in the order the loop's counted waits expect: as if requested in tiles -2 and -1
    if "nowords" not in ABL:
        for ring, ptr, n in ((V_WL, S_PLW, 8), (V_WT, S_PTW, 4)):
            for i in range(n):
                e(f"global_load_dword v{word(ring, i)}, v{V_L4}, s[{ptr}:{ptr + 1}] offset:{i * 256}")
        e("s_waitcnt vmcnt(0)")
        e(f"global_load_dwordx4 {quad(V_WST)}, v{V_L16}, s[{S_PTW}:{S_PTW + 1}] offset:1024")
    raw_request(g_, 0, 3072)
    bump(e, S_PTR, 4096)
    if "nowords" not in ABL:
        e(f"global_load_dwordx4 {quad(V_WSL)}, v{V_L16}, s[{S_PLW}:{S_PLW + 1}] offset:2048")
        bump(e, S_PLW, 3072)       # the loop requests the entering stream's rows 4 ((u + 5) / 2) .. at (u / 2) * 1024 from here
        bump(e, S_PTW, 2048)       # ... the leaving stream's rows 4 ((u + 4) / 2) ..
    raw_request(g_, 1, 0)
    bump(e, S_PTR, (RAW_AHEAD - 4) * 1024)       # the loop's term offsets are (u % 4) * 1024 from here
    # first word of tile 0 -> LO_A, and the look-ups of its first windows = windows 32 .. of a tile u = UNROLL - 1
    e(f"v_alignbit_b32 v{V_FUN}, v{word(V_WL, 1)}, v{word(V_WL, 0)}, s{S_SHL}")
    e(f"v_alignbit_b32 v{V_FUN + 1}, v{word(V_WT, 1)}, v{word(V_WT, 0)}, s{S_SHT}")
    gen_codes(e, "A")
    for x in range(LOOK):
        gen_E(g_, UNROLL - 1, 32 + x)
    for x in range(LOOK):
        gen_R(g_, UNROLL - 1, 32 + x)
    e("LOOP_%=:")
    for u in range(UNROLL):
        gen_tile(g_, u)
    e(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    e(f"s_cmp_lg_u32 s{S_CNT}, 0")
    e("s_cbranch_scc1 LOOP_%=")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    if not bits and "nocapture" not in ABL:
        # the samples still kept (fewer than KEEP): 8 B each
        for k in range(KEEP - 1):
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
    raw_request(g_, 0, 3072)
    bump(e, S_PTR, 4096)
    raw_request(g_, 1, 0)
    bump(e, S_PTR, (RAW_AHEAD - 4) * 1024)
    e("IDLE_LOOP_%=:")
    for u in range(UNROLL):
        gen_idle_tile(g_, u)
    e(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    e(f"s_cmp_lg_u32 s{S_CNT}, 0")
    e("s_cbranch_scc1 IDLE_LOOP_%=")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
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
        f.write(f"#define GARLIC_FEED_WG_PER_CU {4 if (LDS_TOTAL <= 40960 and 142 - _SH <= 128) else 3}   // what the loop's registers and LDS allow\n")
        f.write(f"#define GARLIC_FEED_REACH {(RAW_AHEAD + UNROLL + 4) * 32}   // SNPs past a run's last window the loop may request\n")
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
