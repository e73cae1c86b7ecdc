#!/usr/bin/env python3
"""Generates garlic_amd/csrc/chain_loop_gfx950.inc: the hand-scheduled steady-state loop of the LOD
chain kernel -- ONE inline-asm block executed by the 4 wavefronts of a workgroup, each in its own
role on its own SIMD (gfx950, wave64; one workgroup = one (SNP run, 64 individuals) item at a
time).

Why hand-scheduled, why roles: a single wavefront issues about one instruction every 4 cycles, and
a ds_write_b128 / global_store_dwordx4 blocks it for ~28 / ~38 cycles (tools/ubench/
issue_rates.hip), so one wave doing everything needs ~77 cycles per window start.  hipcc
serialises ds_read -> s_waitcnt -> v_add_f64 per step; here every LDS read is issued an 8-step
batch ahead and all waits are counted.  The wave that carries the sequential FP64 chain sets the
pace of the whole kernel, so everything that is not the chain lives in the other three.

  wave 0  CHAIN  per window ONE address op (byte extract), ONE look-up (ds_read_b128 =
                 {t_out, t_in} of the lane's genotype pair) and the dependent chain
                 acc = (acc - t_out) + t_in (two roundings, reference src/garlic-roh.cpp:98-100);
                 acc -> transpose tile (ds_write_b128 per 2 steps).  No vector-memory instruction
                 at all: a request of its own would queue behind POST's stores in the CU's memory
                 pipeline and stall the chain (measured: +30 %).
  wave 1  POST   transposed write-out of finished tiles: 16 x (ds_read_b128 ->
                 global_store_dwordx4 nt), 4 rows x 256 B per store, row groups past the shard's
                 last individual skipped.  Nothing else: when HBM pushes back this wave blocks at
                 the store issue, and anything else in it adds to the stage time one to one.
  wave 2  PRE    LDS-DMA prefetch: per tile the 2 x 32 raw term rows into the NSLOT-deep TAB ring,
                 and every CHPERIOD tiles one chunk (CHROWS word rows, 1 KB requests) of the item's
                 block-major genotype stream into the WROWS-row genotype ring that BOTH SNP
                 streams read (the leaving stream is the entering one W-1 SNPs later); funnel
                 shift of the words; per step the byte 16 * (4 * g_out + g_in) -> EXP ring.
  wave 3  COMB   per tile the 32 step tables {t_out[g_out], t_in[g_in]} for the 16 genotype pairs
                 (16 ds_read_b64 + 8 ds_write_b128 per lane) -> COMB ring.  This is what lets
                 CHAIN do one look-up per window instead of two (and one address op instead of
                 two); in PRE it made PRE the bottleneck (measured), hence its own wave.

The waves are decoupled by counters in LDS instead of barriers (a barrier per tile makes every
POST stall -- HBM back-pressure comes in bursts -- a CHAIN stall):
    tiles_done    CHAIN -> all        tile k is complete in LDS; its input slots are released
    tiles_stored  POST -> CHAIN       tile buffer k has been read back (may be overwritten)
    exp_ready     PRE -> CHAIN        highest tile whose expanded offsets are in the EXP ring
    comb_ready    COMB -> CHAIN       highest tile whose step tables are in the COMB ring
    tabs_landed   PRE -> COMB         number of tiles whose raw term rows have landed
PRE's only vector-memory operations are its own requests and they retire in issue order, so its
wait is an exact vmcnt(n): the requests of the NFLY youngest tiles may stay in flight
(loads_in_flight()).  PRE expands and COMB combines tile k+ELEAD once CHAIN has finished tile k
(rings of NEXP slots); the raw rows were requested NSLOT tiles ahead.

CHAIN per tile (32 window starts x 64 individuals), software pipeline over 8-step batches g:
    A(g)  byte offsets of the batch's entries: 1 VALU per step (v_and / v_bfe / v_lshr)
    R(g)  8 ds_read_b128 into one of two 32-VGPR buffers
    C(g)  the chain + tile writes, with A(g+2) woven in
  iteration g:  wait R(g) | issue R(g+1) | C(g) (+) A(g+2)

LDS map (bytes; lod_kernels.hpp takes GARLIC_CHAIN_LDS_* from the generated file):
      0  generic-path slot (3072; during the loop: PRE's two 1-KB spreading tables), item word (3072),
         flags (3584 .. 3603)
   4096  COMB ring  NEXP x 8192    per step 16 entries x {t_out, t_in}    (CHAIN: immediate offsets)
      +  EXP ring   NEXP x 2048    32 bytes per lane
      +  TAB ring   NSLOT x 2048   {lead term rows 1024, trail term rows 1024}        (LDS-DMA target)
      +  WORD ring  WROWS x 256    genotype word row w at (w % WROWS)                 (LDS-DMA target)
      +  TILE[NTILE][64 rows x 272 B]
Everything addressed with 16-bit DS immediates stays below 64 KB; LDS-DMA targets may lie above
(measured).  All fixed VGPRs live in v64..v187, reused by the four roles (separate register files).

Environment hooks for experiments: GARLIC_NSLOT, GARLIC_NFLY, GARLIC_NTILE, GARLIC_WROWS,
GARLIC_CHUNK, GARLIC_STORE_FLAGS, GARLIC_LOAD_FLAGS, GARLIC_ABLATE (nodp, chainwrite, nosync,
nodma, nochunk, notab, nopost, poststore, noexpand).
"""
import os


ABL = os.environ.get("GARLIC_ABLATE", "")
STORE_FLAGS = os.environ.get("GARLIC_STORE_FLAGS", "nt")  # non-temporal: the scores are written once and never re-read here (measured -2.7 %)

LOAD_FLAGS = os.environ.get("GARLIC_LOAD_FLAGS", "")
NSLOT = int(os.environ.get("GARLIC_NSLOT", "8"))
NTILE = int(os.environ.get("GARLIC_NTILE", "2"))
NFLY = int(os.environ.get("GARLIC_NFLY", "3"))   # tiles of LDS-DMA requests PRE leaves in flight
assert 3 * NFLY <= 63 and NSLOT - NFLY >= 3 and NSLOT % 2 == 0
WROWS = int(os.environ.get("GARLIC_WROWS", "128"))   # genotype ring: word rows (256 B each), power of 2
WMASK = WROWS * 256 - 1
CH = int(os.environ.get("GARLIC_CHUNK", "4"))        # 1 KB requests per genotype chunk (4 word rows each)
CHROWS, CHPERIOD = 4 * CH, 2 * CH                    # rows per chunk; one chunk every CHPERIOD tiles
assert NSLOT % CHPERIOD == 0
NEXP = 4              # depth of the rings CHAIN reads (combined term table, expanded offsets)
ELEAD = 3             # PRE prepares tile k+ELEAD once CHAIN has finished tile k
assert NSLOT % NEXP == 0 and ELEAD < NEXP and ELEAD <= NSLOT - NFLY and NSLOT % NTILE == 0
FLAGS = 3584          # +0 tiles_done, +4 tiles_stored, +8 exp_ready, +12 comb_ready, +16 tabs_landed
SPREAD_IN, SPREAD_OUT = 0, 1024   # PRE's genotype-spreading tables (inside the generic path's slot, idle during the loop)
# Rings.  What CHAIN reads with immediate offsets (16-bit DS offset field) comes first:
#   COMB  tile t -> slot t % NEXP: per window step one 256-B table of the 16 (leaving genotype,
#         entering genotype) combinations, entry = {t_out, t_in}: ONE look-up per window
#   EXP   tile t -> slot t % NEXP: per lane 32 bytes, byte j = 16 * (4 * g_out + g_in) of step j
#   TAB   tile t -> slot t % NSLOT: raw term rows as fetched (lead 1024 B, trail 1024 B; LDS-DMA)
#   WORD  genotype word row w at (w % WROWS) * 256 (LDS-DMA; targets above 64 KB are fine)
COMB_BASE, COMB_SLOT = 4096, 8192
EXP_BASE, EXP_SLOT = COMB_BASE + NEXP * COMB_SLOT, 2048
TAB_BASE, TAB_SLOT = EXP_BASE + NEXP * EXP_SLOT, 2048
T_LTAB, T_TTAB = 0, 1024
WORD_BASE = TAB_BASE + NSLOT * TAB_SLOT
TILE_BASE = WORD_BASE + WROWS * 256
assert WORD_BASE <= 65535
TILE_BUF = 64 * 34 * 8
TPITCH_B = 34 * 8
LDS_TOTAL = TILE_BASE + NTILE * TILE_BUF

# ---- fixed VGPRs (clobbered by the block; each wave has its own register file)
# Every role is a different wave with its own register file, so the four roles reuse one compact
# range v64..v185: the kernel's register count decides how many workgroups fit a CU.
V_BUF = [64, 96]            # CHAIN: two term buffers, 8 x {t_out, t_in} each
V_ADDR = 128                # CHAIN: 8 LDS byte offsets
V_ACC = 144                 # CHAIN: P0 = [144:145], P1 = [146:147]
V_E = [148, 156]            # CHAIN: two sets of 8 expanded-offset dwords
V_LANE32 = 180              # CHAIN: EXP_BASE + lane * 32
V_TWR = 181                 # CHAIN: tile write address
V_ST = 64                   # POST: 64 VGPRs of store data
V_STOFF = 128               # POST: 16 store offsets
V_TRD, V_TRD2 = 144, 145    # POST: tile-read bases
V_X = 64                    # PRE: 8 expanded dwords being built (+ 8 scratch from V_X + 8)
V_CT, V_CL, V_CW = 101, 102, 103   # PRE: per-lane offsets of the combine step (trail row, lead row, entry)
V_CB = 104                  # PRE: 8 x {t_out, t_in} being combined (32 VGPRs)
V_LL, V_LH, V_TL, V_TH = 84, 85, 86, 87   # PRE: funnel-shifted genotype bits
V_WL1, V_WL2, V_WT1, V_WT2 = 88, 89, 90, 91
V_LANE4, V_LANE16 = 92, 93
V_LADDR, V_TADDR = 94, 95   # PRE: ring byte offset (+ lane*4) of word +0 of the next tile to expand
V_A1L, V_A1T = 96, 97       # PRE: ring offsets of word +1
V_LANE32P = 98              # PRE: EXP_BASE + lane * 32
V_LC, V_TC = 99, 100
V_FLAG, V_TMP0, V_TMP1 = 182, 184, 185   # all roles; v[184:187] = destination of the flag reads
# ---- fixed SGPRs
S_PCHUNK, S_ROFF, S_NCH, S_PLTAB, S_PTTAB = 40, 42, 43, 44, 46
S_OUT = 50
S_CNT = 52
S_SHL, S_SHT = 53, 54
S_TMP = 55
S_K = 56
S_F0, S_F1 = 57, 58
S_ROWS = 59                # POST: valid individuals (rows) of this item

CLOBBER_V = list(range(64, 188))
CLOBBER_S = list(range(40, 60))


class Gen:
    def __init__(self):
        self.out = []
        self.issued = 0
        self.complete = 0

    def emit(self, s):
        self.out.append(s)

    def lds(self, s):
        self.out.append(s)
        self.issued += 1
        return self.issued

    def wait_lds(self, op):
        if op <= self.complete:
            return
        n = min(self.issued - op, 15)
        self.emit(f"s_waitcnt lgkmcnt({n})")
        self.complete = self.issued - n

    def drained(self):
        """the caller just emitted s_waitcnt lgkmcnt(0)"""
        self.issued = self.complete = 0


def pair(r):
    return f"v[{r}:{r + 1}]"


def quad(r):
    return f"v[{r}:{r + 3}]"


# ------------------------------------------------------------------ CHAIN
def addr_ops(j, eset):
    """the VALU op producing the LDS byte offset of step j's {t_out, t_in} entry inside the step's
    256-B table: byte j%4 of dword j//4 of the tile's expanded offsets (register set `eset`)"""
    dst, src, b = V_ADDR + j % 8, V_E[eset] + j // 4, j % 4
    if b == 0:
        return [f"v_and_b32_e32 v{dst}, 0xff, v{src}"]
    if b == 3:
        return [f"v_lshrrev_b32_e32 v{dst}, 24, v{src}"]
    return [f"v_bfe_u32 v{dst}, v{src}, {8 * b}, 8"]


def all_addr_ops(n, eset):
    ops = []
    for i in range(8):
        ops += addr_ops(8 * n + i, eset)
    return ops


def gen_R(g, slot, n):
    """issue the 8 look-ups of batch n of the tile living in COMB/EXP slot `slot`: one ds_read_b128
    per window step = {t_out, t_in} of this lane's genotype pair"""
    buf = V_BUF[n % 2]
    last = 0
    for i in range(8):
        j = 8 * n + i
        last = g.lds(f"ds_read_b128 {quad(buf + 4 * i)}, v{V_ADDR + i} offset:{COMB_BASE + slot * COMB_SLOT + 256 * j}")
    return last


def gen_C(g, n, a_ops, tbuf, write=True):
    """chain of batch n, the next-but-one batch's address ops woven in (1 per step)"""
    buf = V_BUF[n % 2]
    a_ops = list(a_ops)
    P0, P1 = V_ACC, V_ACC + 2
    for i in range(8):
        j = 8 * n + i
        dst, prev = (P0, P1) if j % 2 == 0 else (P1, P0)
        if "nodp" not in ABL:
            g.emit(f"v_add_f64 {pair(dst)}, {pair(prev)}, -{pair(buf + 4 * i)}")
        if a_ops:
            g.emit(a_ops.pop(0))
        if "nodp" not in ABL:
            g.emit(f"v_add_f64 {pair(dst)}, {pair(dst)}, {pair(buf + 4 * i + 2)}")
        if a_ops:
            g.emit(a_ops.pop(0))
        if j % 2 == 1 and write and "chainwrite" not in ABL:
            g.lds(f"ds_write_b128 v{V_TWR}, {quad(V_ACC)} offset:{tbuf * TILE_BUF + 8 * (j - 1)}")
    assert not a_ops


def gen_exp_reads(g, slot, eset):
    """CHAIN: the tile's 8 expanded-offset dwords (this lane's 32 bytes)"""
    base = slot * EXP_SLOT       # V_LANE32 holds EXP_BASE + lane*32
    last = 0
    for h in range(2):
        last = g.lds(f"ds_read_b128 {quad(V_E[eset] + 4 * h)}, v{V_LANE32} offset:{base + 16 * h}")
    return last


def chain_tile(g, slot, uid):
    """CHAIN, one tile k (unrolled position slot = k % NSLOT): COMB/EXP slot k % NEXP, tile buffer k % NTILE"""
    tbuf = slot % NTILE
    nxt = (slot + 1) % NEXP
    slot = slot % NEXP
    e = g.emit
    if "nosync" not in ABL:
        # wait until POST has (a) read back the tile that last used this tile buffer and (b) confirmed
        # the inputs of the NEXT tile (its words are read in batch 1, its terms from batch 3 on)
        e(f"CHAIN_POLL_{uid}_%=:")
        e(f"ds_read_b128 v[{V_TMP0}:{V_TMP0 + 3}], v{V_FLAG}")   # done, stored, exp_ready, comb_ready
        e("s_waitcnt lgkmcnt(0)")
        e(f"v_readfirstlane_b32 s{S_F0}, v{V_TMP0 + 1}")
        e(f"v_readfirstlane_b32 s{S_F1}, v{V_TMP0 + 2}")
        e(f"v_readfirstlane_b32 s{S_TMP}, v{V_TMP0 + 3}")
        e(f"s_add_u32 s{S_F0}, s{S_F0}, {NTILE - 1}")
        e(f"s_cmp_ge_u32 s{S_F0}, s{S_K}")           # tiles_stored >= k - (NTILE-1)
        e(f"s_cbranch_scc0 CHAIN_SLEEP_{uid}_%=")
        e(f"s_min_u32 s{S_F1}, s{S_F1}, s{S_TMP}")   # both the offsets and the tables of tile k+1
        e(f"s_cmp_gt_u32 s{S_F1}, s{S_K}")           # ready >= k + 1
        e(f"s_cbranch_scc1 CHAIN_GO_{uid}_%=")
        e(f"CHAIN_SLEEP_{uid}_%=:")
        e("s_sleep 1")
        e(f"s_branch CHAIN_POLL_{uid}_%=")
        e(f"CHAIN_GO_{uid}_%=:")
    else:
        e("s_waitcnt lgkmcnt(0)")
    g.drained()
    es, en = slot % 2, (slot + 1) % 2   # expanded-offset register set of this / the next tile

    def body(write):
        # batch 0: R(k,0) is complete (the poll drained the LDS queue)
        r1 = gen_R(g, slot, 1)
        gen_C(g, 0, all_addr_ops(2, es), tbuf, write)
        # batch 1: fetch the next tile's expanded offsets into the other register set
        exp = gen_exp_reads(g, nxt, en)
        g.wait_lds(r1)
        r2 = gen_R(g, slot, 2)
        gen_C(g, 1, all_addr_ops(3, es), tbuf, write)
        # batch 2
        g.wait_lds(r2)
        assert exp <= g.complete
        r3 = gen_R(g, slot, 3)
        gen_C(g, 2, all_addr_ops(0, en), tbuf, write)
        # batch 3
        g.wait_lds(r3)
        gen_R(g, nxt, 0)
        gen_C(g, 3, all_addr_ops(1, en), tbuf, write)

    body(True)
    # publish: tile k complete in LDS (its writes are waited for), input slot k released
    e(f"s_add_u32 s{S_K}, s{S_K}, 1")
    e(f"v_mov_b32_e32 v{V_TMP0}, s{S_K}")
    e("s_waitcnt lgkmcnt(0)")
    g.drained()
    e(f"ds_write_b32 v{V_FLAG}, v{V_TMP0}")


def gen_chain(g):
    e = g.emit
    e("ROLE_CHAIN_%=:")
    e(f"s_mov_b32 s{S_CNT}, %[ntiles]")
    e(f"s_mov_b32 s{S_K}, 0")
    e(f"v_mov_b64 {pair(V_ACC + 2)}, %[acc]")
    e(f"v_lshlrev_b32_e32 v{V_LANE32}, 5, %[lane]")
    e(f"v_add_u32_e32 v{V_LANE32}, {EXP_BASE}, v{V_LANE32}")
    e(f"v_mul_u32_u24_e32 v{V_TWR}, {TPITCH_B}, %[lane]")
    e(f"v_add_u32_e32 v{V_TWR}, {TILE_BASE}, v{V_TWR}")
    e(f"v_mov_b32_e32 v{V_FLAG}, {FLAGS}")
    e(f"v_mov_b32_e32 v{V_TMP0}, 0")
    e(f"v_mov_b32_e32 v{V_TMP1}, 0")
    e(f"ds_write_b32 v{V_FLAG}, v{V_TMP0}")                                  # tiles_done = 0
    e(f"ds_write2_b32 v{V_FLAG}, v{V_TMP0}, v{V_TMP1} offset0:1 offset1:2")  # stored = exp_ready = 0
    e(f"ds_write2_b32 v{V_FLAG}, v{V_TMP0}, v{V_TMP1} offset0:3 offset1:4")  # comb_ready = tabs_landed = 0
    e("s_waitcnt lgkmcnt(0)")
    e("s_barrier")  # the previous item's counters are gone; POST may start
    e("CHAIN_FIRST_%=:")  # wait for the rings' initial fill
    e(f"ds_read2_b32 v[{V_TMP0}:{V_TMP1}], v{V_FLAG} offset0:2 offset1:3")
    e("s_waitcnt lgkmcnt(0)")
    e(f"v_readfirstlane_b32 s{S_F0}, v{V_TMP0}")
    e(f"v_readfirstlane_b32 s{S_F1}, v{V_TMP1}")
    e(f"s_min_u32 s{S_F1}, s{S_F1}, s{S_F0}")
    e(f"s_cmp_gt_u32 s{S_F1}, 0")
    e("s_cbranch_scc1 CHAIN_START_%=")
    e("s_sleep 2")
    e("s_branch CHAIN_FIRST_%=")
    e("CHAIN_START_%=:")
    g.drained()
    # pipeline prologue for the first tile (slot 0, register set 0)
    x = gen_exp_reads(g, 0, 0)
    g.wait_lds(x)
    for op in all_addr_ops(0, 0):
        e(op)
    gen_R(g, 0, 0)
    for op in all_addr_ops(1, 0):
        e(op)
    e("CHAIN_LOOP_%=:")
    for slot in range(NSLOT):
        chain_tile(g, slot, slot)
        e(f"s_cmp_eq_u32 s{S_K}, s{S_CNT}")
        if slot < NSLOT - 1:
            e("s_cbranch_scc1 CHAIN_DONE_%=")
        else:
            e("s_cbranch_scc0 CHAIN_LOOP_%=")
    e("CHAIN_DONE_%=:")
    e("s_waitcnt lgkmcnt(0)")
    e(f"v_mov_b64 %[acc], {pair(V_ACC + 2)}")
    e("s_branch DONE_%=")


# ------------------------------------------------------------------ PRE helpers
def gen_chunk(g):
    """one 1 KB LDS-DMA request: the next 4 word rows of the item's genotype stream into the
    genotype ring.  An SALU write of M0 needs a wait state before the LDS-DMA reads it."""
    e = g.emit
    e(f"s_add_u32 m0, s{S_ROFF}, {WORD_BASE}")
    e("s_nop 0")
    e((f"global_load_lds_dwordx4 v{V_LANE16}, s[{S_PCHUNK}:{S_PCHUNK + 1}] " + LOAD_FLAGS).rstrip())
    e(f"s_add_u32 s{S_PCHUNK}, s{S_PCHUNK}, 1024")
    e(f"s_addc_u32 s{S_PCHUNK + 1}, s{S_PCHUNK + 1}, 0")
    e(f"s_add_u32 s{S_ROFF}, s{S_ROFF}, 1024")
    e(f"s_and_b32 s{S_ROFF}, s{S_ROFF}, {WMASK}")


def gen_prefetch(g, slot, chunk):
    """requests for one tile into ring slot `slot`: its 2 x 32 term rows, and (every other tile)
    the next genotype chunk"""
    if "nodma" in ABL:
        return
    if chunk and "nochunk" not in ABL:
        for _ in range(CH):
            gen_chunk(g)
    tb = TAB_BASE + slot * TAB_SLOT
    if "notab" in ABL:
        return
    for m0, ptr in ((tb + T_LTAB, S_PLTAB), (tb + T_TTAB, S_PTTAB)):
        g.emit(f"s_mov_b32 m0, {m0}")
        g.emit("s_nop 0")
        g.emit((f"global_load_lds_dwordx4 v{V_LANE16}, s[{ptr}:{ptr + 1}] " + LOAD_FLAGS).rstrip())
        g.emit(f"s_add_u32 s{ptr}, s{ptr}, 1024")
        g.emit(f"s_addc_u32 s{ptr + 1}, s{ptr + 1}, 0")


def loads_in_flight(slot):
    """PRE's vector-memory operations retire in issue order: after iteration `slot` issued its
    requests, the NFLY youngest iterations' requests may stay in flight (2 per tile + 1 chunk
    per even tile)"""
    ntab = 0 if "notab" in ABL else 2
    nch = 0 if "nochunk" in ABL else CH
    return sum(ntab + (nch if ((slot - d) % NSLOT) % CHPERIOD == 0 else 0) for d in range(NFLY))


def post_expand(g, slot):
    """PRE: prepare the tile living in TAB slot `slot` (unrolled position = tile % NSLOT) for CHAIN.
    (1) Genotype words (+1, +2; word +0 is carried) of both SNP streams from the genotype ring,
        funnel-shifted to the tile's first step; per step the pair (leaving genotype g_out,
        entering genotype g_in) becomes one byte 16 * (4 * g_out + g_in) = the offset of the
        pair's entry in the step's table; this lane's 32 bytes go to the EXP slot.
    (2) The tile's 32 step tables: entry (g_out, g_in) = {t_out[g_out], t_in[g_in]} from the raw
        term rows, 16 entries x 16 B per step, 8 entries per lane -> COMB slot.  CHAIN then needs
        ONE address op and ONE ds_read_b128 per window instead of two and two."""
    if "noexpand" in ABL:
        return
    e = g.emit
    s4 = slot % NEXP
    for a1, a0 in ((V_A1L, V_LADDR), (V_A1T, V_TADDR)):
        e(f"v_add_u32_e32 v{a1}, 0x100, v{a0}")
        e(f"v_and_b32_e32 v{a1}, {WMASK}, v{a1}")
        e(f"v_add_u32_e32 v{a0}, 0x100, v{a1}")
        e(f"v_and_b32_e32 v{a0}, {WMASK}, v{a0}")
    g.lds(f"ds_read_b32 v{V_WL1}, v{V_A1L} offset:{WORD_BASE}")
    g.lds(f"ds_read_b32 v{V_WL2}, v{V_LADDR} offset:{WORD_BASE}")
    g.lds(f"ds_read_b32 v{V_WT1}, v{V_A1T} offset:{WORD_BASE}")
    w = g.lds(f"ds_read_b32 v{V_WT2}, v{V_TADDR} offset:{WORD_BASE}")
    g.wait_lds(w)
    e(f"v_alignbit_b32 v{V_LL}, v{V_WL1}, v{V_LC}, s{S_SHL}")
    e(f"v_alignbit_b32 v{V_LH}, v{V_WL2}, v{V_WL1}, s{S_SHL}")
    e(f"v_alignbit_b32 v{V_TL}, v{V_WT1}, v{V_TC}, s{S_SHT}")
    e(f"v_alignbit_b32 v{V_TH}, v{V_WT2}, v{V_WT1}, s{S_SHT}")
    e(f"v_mov_b32_e32 v{V_LC}, v{V_WL2}")
    e(f"v_mov_b32_e32 v{V_TC}, v{V_WT2}")
    if "bfeexpand" in ABL:                        # the all-VALU expansion this replaced: 16 instructions per dword
        t = [V_X + 8 + i for i in range(8)]      # scratch: g_in 0-3, g_out 4-7
        for d in range(8):                        # dword d holds steps 4d..4d+3
            lsrc, tsrc = (V_LL, V_TL) if d < 4 else (V_LH, V_TH)
            o = 8 * (d % 4)
            dst = V_X + d
            for i in range(4):
                e(f"v_bfe_u32 v{t[i]}, v{lsrc}, {o + 2 * i}, 2")
                e(f"v_bfe_u32 v{t[4 + i]}, v{tsrc}, {o + 2 * i}, 2")
            for i in range(4):
                e(f"v_lshl_or_b32 v{t[i]}, v{t[4 + i]}, 2, v{t[i]}")      # 4 * g_out + g_in
            e(f"v_lshl_or_b32 v{t[0]}, v{t[1]}, 8, v{t[0]}")
            e(f"v_lshl_or_b32 v{t[2]}, v{t[3]}, 8, v{t[2]}")
            e(f"v_lshl_or_b32 v{dst}, v{t[2]}, 16, v{t[0]}")
            e(f"v_lshlrev_b32_e32 v{dst}, 4, v{dst}")
    else:
        # Four genotypes (one byte of a stream) -> four offset bytes through two 256-entry tables in LDS
        # (SPREAD_IN[b] = genotype k of b, times 16, in byte k; SPREAD_OUT the same times 64; written by
        # lod_chain_kernel before the loop): per dword two byte extractions (SDWA, scaled to a table
        # offset), two look-ups and an OR instead of 16 VALU instructions -- the expansion was what PRE
        # spent most of its time on, and in the thinned kernel PRE sets the pace.
        sel = "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD"
        for d in range(8):
            lsrc, tsrc = (V_LL, V_TL) if d < 4 else (V_LH, V_TH)
            e(f"v_lshlrev_b32_sdwa v{V_X + d}, 2, v{lsrc} {sel} src1_sel:BYTE_{d % 4}")
            e(f"v_lshlrev_b32_sdwa v{V_X + 8 + d}, 2, v{tsrc} {sel} src1_sel:BYTE_{d % 4}")
        rd = []
        for d in range(8):
            g.lds(f"ds_read_b32 v{V_X + d}, v{V_X + d} offset:{SPREAD_IN}")
            rd.append(g.lds(f"ds_read_b32 v{V_X + 8 + d}, v{V_X + 8 + d} offset:{SPREAD_OUT}"))
        for d in range(8):
            g.wait_lds(rd[d])
            e(f"v_or_b32_e32 v{V_X + d}, v{V_X + d}, v{V_X + 8 + d}")
    for h in range(2):                            # V_LANE32P holds EXP_BASE + lane*32
        g.lds(f"ds_write_b128 v{V_LANE32P}, {quad(V_X + 4 * h)} offset:{s4 * EXP_SLOT + 16 * h}")


def comb_build(g, slot):
    """COMB wave: the 32 step tables of the tile whose raw term rows live in TAB slot `slot`:
    entry (g_out, g_in) = {t_out[g_out], t_in[g_in]}, 16 entries x 16 B per step, 8 entries per
    lane (lane -> step (lane >> 4) + 4*kk, entry lane & 15)."""
    s4 = slot % NEXP
    tb = TAB_BASE + slot * TAB_SLOT
    rd = []
    for kk in range(8):
        g.lds(f"ds_read_b64 {pair(V_CB + 4 * kk)}, v{V_CT} offset:{tb + T_TTAB + 128 * kk}")
        rd.append(g.lds(f"ds_read_b64 {pair(V_CB + 4 * kk + 2)}, v{V_CL} offset:{tb + T_LTAB + 128 * kk}"))
    for kk in range(8):                           # V_CW holds lane*16: entry e = lane + 64*kk
        g.wait_lds(rd[kk])
        g.lds(f"ds_write_b128 v{V_CW}, {quad(V_CB + 4 * kk)} offset:{COMB_BASE + s4 * COMB_SLOT + 1024 * kk}")


# ------------------------------------------------------------------ POST
def post_tile(g, slot, uid):
    """POST, one tile k (tile buffer = k % NTILE): transposed write-out.  Its stores are never waited
    for; when HBM pushes back the wave simply blocks at the store issue (which is why nothing else
    lives in this wave)."""
    tbuf = slot % NTILE
    e = g.emit
    e(f"POST_POLL_{uid}_%=:")  # wait for CHAIN to finish tile k
    e(f"ds_read_b32 v{V_TMP0}, v{V_FLAG}")
    e("s_waitcnt lgkmcnt(0)")
    e(f"v_readfirstlane_b32 s{S_F0}, v{V_TMP0}")
    e(f"s_cmp_gt_u32 s{S_F0}, s{S_K}")               # tiles_done >= k + 1
    e(f"s_cbranch_scc1 POST_GO_{uid}_%=")
    e("s_sleep 1")
    e(f"s_branch POST_POLL_{uid}_%=")
    e(f"POST_GO_{uid}_%=:")
    g.drained()
    if "nopost" not in ABL:
        ids = []
        for q in range(16):
            # DS offsets are 16 bit: tile buffers 2,3 are addressed from a second base register
            base, rel = (V_TRD, tbuf) if tbuf < 2 else (V_TRD2, tbuf - 2)
            ids.append(g.lds(f"ds_read_b128 {quad(V_ST + 4 * q)}, v{base} offset:{rel * TILE_BUF + q * 4 * TPITCH_B}"))
        for q in range(16):
            g.wait_lds(ids[q])
            if "poststore" not in ABL:
                # row groups wholly past the shard's last individual are pad rows: not stored
                e(f"s_cmp_gt_u32 s{S_ROWS}, {4 * q}")
                e(f"s_cbranch_scc0 POST_SKIP_{uid}_{q}_%=")
                e(f"global_store_dwordx4 v{V_STOFF + q}, {quad(V_ST + 4 * q)}, s[{S_OUT}:{S_OUT + 1}] {STORE_FLAGS}".rstrip())
                e(f"POST_SKIP_{uid}_{q}_%=:")
        e(f"s_add_u32 s{S_OUT}, s{S_OUT}, 256")
        e(f"s_addc_u32 s{S_OUT + 1}, s{S_OUT + 1}, 0")
    assert g.complete == g.issued
    # publish: tile buffer k has been read back
    e(f"s_add_u32 s{S_K}, s{S_K}, 1")
    e(f"v_mov_b32_e32 v{V_TMP0}, s{S_K}")
    e(f"ds_write_b32 v{V_FLAG}, v{V_TMP0} offset:4")


def gen_post(g):
    e = g.emit
    e("ROLE_POST_%=:")
    e(f"s_mov_b32 s{S_ROWS}, %[rows]")
    e(f"s_mov_b64 s[{S_OUT}:{S_OUT + 1}], %[out]")
    e(f"s_mov_b32 s{S_CNT}, %[ntiles]")
    e(f"s_mov_b32 s{S_K}, 0")
    e(f"v_mov_b32_e32 v{V_FLAG}, {FLAGS}")
    # tile read address (lane>>4)*272 + (lane&15)*16 ; store offset (lane>>4)*pitch8 + (lane&15)*16
    e(f"v_lshrrev_b32_e32 v{V_TRD}, 4, %[lane]")
    e(f"v_and_b32_e32 v{V_STOFF}, 15, %[lane]")
    e(f"v_lshlrev_b32_e32 v{V_STOFF}, 4, v{V_STOFF}")
    e(f"v_mul_u32_u24_e32 v{V_STOFF + 1}, {TPITCH_B}, v{V_TRD}")
    e(f"v_add_u32_e32 v{V_STOFF + 1}, v{V_STOFF + 1}, v{V_STOFF}")
    e(f"v_mul_lo_u32 v{V_TRD}, v{V_TRD}, %[pitch8]")
    e(f"v_add_u32_e32 v{V_STOFF}, v{V_TRD}, v{V_STOFF}")
    e(f"v_add_u32_e32 v{V_TRD}, {TILE_BASE}, v{V_STOFF + 1}")
    e(f"v_add_u32_e32 v{V_TRD2}, {2 * TILE_BUF}, v{V_TRD}")
    e(f"s_lshl_b32 s{S_TMP}, %[pitch8], 2")
    for q in range(1, 16):
        e(f"v_add_u32_e32 v{V_STOFF + q}, s{S_TMP}, v{V_STOFF + q - 1}")
    e("s_barrier")  # CHAIN has reset the counters
    e("POST_LOOP_%=:")
    for idx in range(NSLOT):
        post_tile(g, idx, idx)
        e(f"s_cmp_eq_u32 s{S_K}, s{S_CNT}")
        if idx < NSLOT - 1:
            e("s_cbranch_scc1 POST_DONE_%=")
        else:
            e("s_cbranch_scc0 POST_LOOP_%=")
    e("POST_DONE_%=:")
    e("s_waitcnt lgkmcnt(0)")
    e("s_branch DONE_%=")


# ------------------------------------------------------------------ PRE
def pre_tile(g, slot, uid):
    """PRE, iteration k: once CHAIN has finished tile k its TAB slot is free -> request tile
    k+NSLOT into it; the rows requested NFLY iterations ago (tile k+NSLOT-NFLY) have landed ->
    tell COMB; expand tile k+ELEAD."""
    e = g.emit
    e(f"PRE_POLL_{uid}_%=:")
    e(f"ds_read_b32 v{V_TMP0}, v{V_FLAG}")
    e("s_waitcnt lgkmcnt(0)")
    e(f"v_readfirstlane_b32 s{S_F0}, v{V_TMP0}")
    e(f"s_cmp_gt_u32 s{S_F0}, s{S_K}")               # tiles_done >= k + 1
    e(f"s_cbranch_scc1 PRE_GO_{uid}_%=")
    e("s_sleep 1")
    e(f"s_branch PRE_POLL_{uid}_%=")
    e(f"PRE_GO_{uid}_%=:")
    g.drained()
    if "prefirst" in ABL:
        gen_prefetch(g, slot, slot % CHPERIOD == 0)
        e(f"s_waitcnt vmcnt({loads_in_flight(slot)})")
        e(f"s_add_u32 s{S_F1}, s{S_K}, {NSLOT - NFLY + 1}")
        e(f"s_max_u32 s{S_F1}, s{S_F1}, {NSLOT}")
        e(f"v_mov_b32_e32 v{V_TMP1}, s{S_F1}")
        e(f"ds_write_b32 v{V_FLAG}, v{V_TMP1} offset:16")  # tabs_landed = tiles 0 .. k+NSLOT-NFLY (a count)
    # what CHAIN waits for comes first: the requests below may block at issue behind POST's stores
    post_expand(g, (slot + ELEAD) % NSLOT)
    e(f"s_add_u32 s{S_F1}, s{S_K}, {ELEAD}")
    e(f"v_mov_b32_e32 v{V_TMP1}, s{S_F1}")
    e("s_waitcnt lgkmcnt(0)")
    g.drained()
    e(f"ds_write_b32 v{V_FLAG}, v{V_TMP1} offset:8")   # inputs_ready = k + ELEAD
    if "prefirst" not in ABL:
        gen_prefetch(g, slot, slot % CHPERIOD == 0)
        e(f"s_waitcnt vmcnt({loads_in_flight(slot)})")
        e(f"s_add_u32 s{S_F1}, s{S_K}, {NSLOT - NFLY + 1}")
        e(f"s_max_u32 s{S_F1}, s{S_F1}, {NSLOT}")
        e(f"v_mov_b32_e32 v{V_TMP1}, s{S_F1}")
        e(f"ds_write_b32 v{V_FLAG}, v{V_TMP1} offset:16")  # tabs_landed = tiles 0 .. k+NSLOT-NFLY (a count)
    e(f"s_add_u32 s{S_K}, s{S_K}, 1")


def gen_pre(g):
    e = g.emit
    e("ROLE_PRE_%=:")
    e(f"s_mov_b64 s[{S_PCHUNK}:{S_PCHUNK + 1}], %[pchunk]")
    e(f"s_mov_b32 s{S_ROFF}, %[roff0]")
    e(f"s_mov_b32 s{S_NCH}, %[nchunk0]")
    e(f"s_mov_b64 s[{S_PLTAB}:{S_PLTAB + 1}], %[pltab]")
    e(f"s_mov_b64 s[{S_PTTAB}:{S_PTTAB + 1}], %[pttab]")
    e(f"s_mov_b32 s{S_CNT}, %[ntiles]")
    e(f"s_mov_b32 s{S_K}, 0")
    e(f"s_mov_b32 s{S_SHL}, %[shl]")
    e(f"s_mov_b32 s{S_SHT}, %[sht]")
    e(f"v_mov_b32_e32 v{V_LC}, %[lc]")
    e(f"v_mov_b32_e32 v{V_TC}, %[tc]")
    e(f"v_mov_b32_e32 v{V_FLAG}, {FLAGS}")
    e(f"v_lshlrev_b32_e32 v{V_LANE4}, 2, %[lane]")
    e(f"v_lshlrev_b32_e32 v{V_LANE16}, 4, %[lane]")
    e(f"v_lshlrev_b32_e32 v{V_LANE32P}, 5, %[lane]")
    e(f"v_add_u32_e32 v{V_LANE32P}, {EXP_BASE}, v{V_LANE32P}")
    e(f"v_add_u32_e32 v{V_LADDR}, %[laddr0], v{V_LANE4}")
    e(f"v_add_u32_e32 v{V_TADDR}, %[taddr0], v{V_LANE4}")
    e("s_barrier")  # CHAIN has reset the counters
    if "nodma" not in ABL:
        e("PRE_FILL_%=:")   # genotype ring: the chunks covering both streams' first NSLOT tiles
        gen_chunk(g)
        e(f"s_sub_u32 s{S_NCH}, s{S_NCH}, 1")
        e(f"s_cmp_lg_u32 s{S_NCH}, 0")
        e("s_cbranch_scc1 PRE_FILL_%=")
    for slot in range(NSLOT):  # term rows of tiles 0..NSLOT-1
        gen_prefetch(g, slot, False)
    e("s_waitcnt vmcnt(0)")
    g.drained()
    e(f"v_mov_b32_e32 v{V_TMP1}, {NSLOT}")
    e(f"ds_write_b32 v{V_FLAG}, v{V_TMP1} offset:16")      # tabs_landed: NSLOT tiles (0 = none yet)
    for slot in range(ELEAD):  # tiles 0..ELEAD-1 prepared up front; the loop stays ELEAD ahead
        post_expand(g, slot)
    e("s_waitcnt lgkmcnt(0)")
    g.drained()
    e(f"v_mov_b32_e32 v{V_TMP0}, {ELEAD - 1}")
    e(f"ds_write_b32 v{V_FLAG}, v{V_TMP0} offset:8")
    e("PRE_LOOP_%=:")
    for idx in range(NSLOT):
        pre_tile(g, idx, idx)
        e(f"s_cmp_eq_u32 s{S_K}, s{S_CNT}")
        if idx < NSLOT - 1:
            e("s_cbranch_scc1 PRE_DONE_%=")
        else:
            e("s_cbranch_scc0 PRE_LOOP_%=")
    e("PRE_DONE_%=:")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")  # run-ahead requests land before the ring is reused
    e("s_branch DONE_%=")


# ------------------------------------------------------------------ COMB
def comb_tile(g, slot, uid):
    """COMB, tile t (unrolled position t % NSLOT): wait for its raw term rows (PRE's tabs_landed)
    and for the COMB slot (tile t - NEXP consumed by CHAIN), build, publish comb_ready = t"""
    e = g.emit
    e(f"COMB_POLL_{uid}_%=:")
    e(f"ds_read2_b32 v[{V_TMP0}:{V_TMP1}], v{V_FLAG} offset0:0 offset1:4")   # tiles_done, tabs_landed
    e("s_waitcnt lgkmcnt(0)")
    e(f"v_readfirstlane_b32 s{S_F0}, v{V_TMP0}")
    e(f"v_readfirstlane_b32 s{S_F1}, v{V_TMP1}")
    e(f"s_add_u32 s{S_F0}, s{S_F0}, {NEXP - 1}")
    e(f"s_cmp_ge_u32 s{S_F0}, s{S_K}")               # tiles_done >= t - (NEXP - 1)
    e(f"s_cbranch_scc0 COMB_SLEEP_{uid}_%=")
    e(f"s_cmp_gt_u32 s{S_F1}, s{S_K}")               # tabs_landed (a count) > t
    e(f"s_cbranch_scc1 COMB_GO_{uid}_%=")
    e(f"COMB_SLEEP_{uid}_%=:")
    e("s_sleep 1")
    e(f"s_branch COMB_POLL_{uid}_%=")
    e(f"COMB_GO_{uid}_%=:")
    g.drained()
    comb_build(g, slot)
    e(f"v_mov_b32_e32 v{V_TMP1}, s{S_K}")
    e("s_waitcnt lgkmcnt(0)")
    g.drained()
    e(f"ds_write_b32 v{V_FLAG}, v{V_TMP1} offset:12")  # comb_ready = t
    e(f"s_add_u32 s{S_K}, s{S_K}, 1")


def gen_comb(g):
    e = g.emit
    e("ROLE_COMB_%=:")
    e(f"s_mov_b32 s{S_CNT}, %[ntiles]")
    e(f"s_mov_b32 s{S_K}, 0")
    e(f"v_mov_b32_e32 v{V_FLAG}, {FLAGS}")
    # lane -> step (lane >> 4) + 4*kk, entry lane & 15 = 4 * g_out + g_in
    e(f"v_lshrrev_b32_e32 v{V_CW}, 4, %[lane]")
    e(f"v_lshlrev_b32_e32 v{V_CW}, 5, v{V_CW}")                 # (lane >> 4) * 32: the step's term row
    e(f"v_bfe_u32 v{V_CT}, %[lane], 2, 2")
    e(f"v_lshl_add_u32 v{V_CT}, v{V_CT}, 3, v{V_CW}")           # + g_out * 8
    e(f"v_and_b32_e32 v{V_CL}, 3, %[lane]")
    e(f"v_lshl_add_u32 v{V_CL}, v{V_CL}, 3, v{V_CW}")           # + g_in * 8
    e(f"v_lshlrev_b32_e32 v{V_CW}, 4, %[lane]")                 # entry address lane * 16
    e("s_barrier")  # CHAIN has reset the counters
    e("COMB_LOOP_%=:")   # tiles 0 .. ntiles (CHAIN wants tile k+1 ready before it starts tile k)
    for idx in range(NSLOT):
        comb_tile(g, idx, idx)
        e(f"s_cmp_gt_u32 s{S_K}, s{S_CNT}")
        if idx < NSLOT - 1:
            e("s_cbranch_scc1 COMB_DONE_%=")
        else:
            e("s_cbranch_scc0 COMB_LOOP_%=")
    e("COMB_DONE_%=:")
    e("s_waitcnt lgkmcnt(0)")
    e("s_branch DONE_%=")


def gen_all():
    g = Gen()
    e = g.emit
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e("s_cmp_eq_u32 %[wave], 1")
    e("s_cbranch_scc1 ROLE_POST_%=")
    e("s_cmp_eq_u32 %[wave], 2")
    e("s_cbranch_scc1 ROLE_PRE_%=")
    e("s_cmp_eq_u32 %[wave], 3")
    e("s_cbranch_scc1 ROLE_COMB_%=")
    gen_chain(g)
    gen_post(g)
    gen_pre(g)
    gen_comb(g)
    e("DONE_%=:")
    e("s_waitcnt lgkmcnt(0)")
    return g.out


def main():
    lines = gen_all()
    here = os.path.dirname(os.path.abspath(__file__))
    # GARLIC_GEN_OUT: write somewhere else (tests/test_generated_cpu.py checks the committed file is current)
    path = os.path.join(os.environ.get("GARLIC_GEN_OUT") or os.path.join(here, "..", "garlic_amd", "csrc"), "chain_loop_gfx950.inc")
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_chain_asm.py -- do not edit; see that file for roles and schedule.\n")
        f.write("// One inline-asm block: steady-state loop of lod_chain_kernel, 4 waves in 4 roles (gfx950).\n")
        f.write(f"#define GARLIC_CHAIN_LDS_TOTAL {LDS_TOTAL}\n")
        f.write(f"#define GARLIC_CHAIN_LDS_TILE0 {TILE_BASE}\n")
        f.write(f"#define GARLIC_CHAIN_NSLOT {NSLOT}\n")
        f.write(f"#define GARLIC_CHAIN_WROWS {WROWS}\n")
        # entering-stream word row minus leaving-stream word row the genotype ring can span
        f.write(f"#define GARLIC_CHAIN_MAX_DW {WROWS - 2 * NSLOT - 3 * CHROWS}\n")
        f.write(f"#define GARLIC_CHAIN_CHROWS {CHROWS}\n")
        f.write(f"#define GARLIC_CHAIN_SPREAD_IN {SPREAD_IN}\n#define GARLIC_CHAIN_SPREAD_OUT {SPREAD_OUT}\n")
        for name, body in (("GARLIC_CHAIN_LOOP_ASM", lines),):
            f.write(f"#define {name} \\\n")
            for ln in body:
                if ln.startswith(";"):
                    continue
                f.write('    "%s\\n\\t" \\\n' % ln)
            f.write('    ""\n')
        f.write("#define GARLIC_CHAIN_LOOP_CLOBBERS \\\n    ")
        regs = ['"v%d"' % r for r in CLOBBER_V] + ['"s%d"' % r for r in CLOBBER_S]
        regs += ['"memory"', '"scc"', '"vcc"']
        chunks = [", ".join(regs[i:i + 12]) for i in range(0, len(regs), 12)]
        f.write(", \\\n    ".join(chunks) + "\n")
    n_instr = sum(1 for ln in lines if not ln.startswith(";") and not ln.endswith(":"))
    print(f"wrote {os.path.normpath(path)}: {n_instr} instructions, LDS {LDS_TOTAL} B, tile0 at {TILE_BASE}")


if __name__ == "__main__":
    main()
