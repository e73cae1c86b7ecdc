#!/usr/bin/env python3
"""Generates garlic_amd/csrc/wlod_loop_gfx950.inc: the whole ordered-sum loop of the tuned wLOD
kernel (wlod_group in variant_kernels.hpp) -- 16 consecutive windows x 64 individuals per wave --
as one inline-asm block for gfx950.

Why by hand: the loop is FP64-VALU bound (per SNP and wave 16 x {v_mul_f64, v_add_f64}: the product
is rounded, then added, garlic-roh.cpp:262-268), but its 16 weights per SNP come through the scalar
cache, and hipcc waits for a scalar load right where it issues it: every SNP step then costs one or
two L2 round trips per wave (measured: 0.42 of the arithmetic rate at 8 waves per SIMD, 0.20 at
W = 400); asked to prefetch in source, it spills SGPRs lane by lane or falls back to vector loads.
Here the weights (2 x s_load_dwordx16) and the score look-up (ds_read_b64) of step i+1 are issued
BEFORE the arithmetic of step i and waited for after it; scalar loads return out of order, so the
wait is lgkmcnt(0).  Ping-pong SGPR tuples / score registers by step parity, no moves.

Step i (SNP s+i, i = 0 .. W+14) serves window r (start s+r) iff 0 <= i-r < W, with weight
D[s+i][i-r] = element 15-r of the 16 contiguous doubles at  Dp_i = D + (s+i)*W + i - 15:
    i < 15        windows 0..i      (unrolled; elements of absent windows are the previous row's
                                     tail: loaded, never used)
    15 <= i < W   all 16 windows    (loop, two steps per iteration)
    i >= W        windows i-W+1..15 (unrolled, in two copies: the parity of W decides which tuple
                                     the first of them finds its weights in)
Per step, besides the arithmetic:
    Dp += (W + 1) * 8;  N <- s_load x2 (Dp)                       next step's weights
    t = (bfe(word, bit, 2) << 3) + row;  scN <- ds_read_b64(t)   next step's score
    row += 32; bit += 2; if bit == 32: word = nextw, nextw = *gaddr, gaddr += 256, bit = 0
    s_waitcnt lgkmcnt(0)
Vector registers and the loop's scalar state are asm operands (the compiler allocates them); the
two weight tuples A = s[36:67], B = s[68:99] and the weight pointer s[34:35] are fixed SGPRs
declared as clobbers (inline asm cannot name a sub-register of a 512-bit operand).
"""
import os

ABL = os.environ.get("GARLIC_WLOD_ABLATE", "")   # experiments: nosload, nolds, nowait, noint (timing only, results wrong)
# Two-block loop: the weights of step i + PFW are touched by plain vector loads (one dword per 64-B line, result
# never read) so that the scalar loads of that step find their lines in L2: the weight table is streamed (every
# weight is used once per pair of blocks), its first reader otherwise waits for HBM with nothing but the other
# waves of the SIMD to cover it.  2M x 1280: W=50 11.6 -> 10.5 ms, W=100 18.2 -> 17.6, W=200 32.9 -> 32.2,
# W=400 62.7 -> 64.3 (close to the FP64 bound the extra vector-memory instructions cost more than they save:
# the wrapper switches the touches off through VCC for W > PFW_MAX_W).
PFW = int(os.environ.get("GARLIC_WLOD_PFW", "4"))
PFW_LOADS = int(os.environ.get("GARLIC_WLOD_PFW_LOADS", "2"))
PFW_MAX_W = 300
R = 16
BASE = {"A": 36, "B": 68}
S_DP = 34


def weight(t, r):
    """SGPR pair of window r's weight: element 15 - r of the 16 contiguous doubles of tuple t"""
    lo = BASE[t] + 2 * (15 - r)
    return f"s[{lo}:{lo + 1}]"


def tup(t, half):
    lo = BASE[t] + 16 * half
    return f"s[{lo}:{lo + 15}]"


def regs_of(parity):
    """(current tuple, next tuple, current score, next score) of a step with this index parity"""
    return ("A", "B", "sc", "scn") if parity == 0 else ("B", "A", "scn", "sc")


# GL variant (per-genotype likelihoods): the score of (SNP, lane) is not a look-up by genotype but the
# lane's own entry of the scaled TGLS term matrix, 512 B per SNP and 64-individual block.  The wave
# keeps a ring of GL_RING rows in LDS, filled by LDS-DMA two rows (1 KB) per request GL_AHEAD rows
# ahead of the step that reads them; LDS-DMA requests retire in order, so the waits are exact
# vmcnt counts.  Operands instead of the genotype ones: [lane8b] = ring base + lane * 8 (VGPR),
# [voff16] = lane * 16 (VGPR, advanced by 1 KB per request), [trow] = address of the block's row of SNP s (SGPR pair),
# [rd] / [wr] = ring byte offsets of the next row to read / to fill (SGPRs), [rbase] = ring base.
GL_RING = int(os.environ.get("GARLIC_WLOD_GL_RING", "8"))   # rows; the request for rows i+GL_AHEAD, +1 overwrites rows i-2, i-1
GL_AHEAD = GL_RING - 2
GL_MASK = GL_RING * 512 - 1


class Gen:
    def __init__(self, gl=False, nb=1):
        self.out = []
        self.uid = 0
        self.gl = gl
        self.nb = nb          # 64-individual blocks per wave (2: every weight serves two blocks)

    def gl_request(self):
        """LDS-DMA of the next two term rows into the ring"""
        e = self.e
        e("s_add_u32 m0, %[rbase], %[wr]")
        e("s_add_u32 %[wr], %[wr], 1024")
        e(f"s_and_b32 %[wr], %[wr], {GL_MASK}")              # (also the wait state M0 needs before the DMA)
        if "nodma" not in ABL:
            e("global_load_lds_dwordx4 %[voff16], %[trow]")
        e("v_add_u32_e32 %[voff16], 0x400, %[voff16]")

    def gl_read(self, dst, parity):
        """score of the next step (row index of that step has the opposite parity of `parity`)"""
        e = self.e
        # requests issued after the one that brought the row: GL_AHEAD/2 when this step has just
        # issued one (even steps), one less otherwise
        assert GL_AHEAD // 2 <= 63           # (counted waits: the VM counter holds 63)
        e(f"s_waitcnt vmcnt({GL_AHEAD // 2 if parity == 0 else GL_AHEAD // 2 - 1})")
        e("v_add_u32_e32 %[vt], %[rd], %[lane8b]")
        e(f"ds_read_b64 %[{dst}], %[vt]")
        e("s_add_u32 %[rd], %[rd], 512")
        e(f"s_and_b32 %[rd], %[rd], {GL_MASK}")

    def e(self, s):
        self.out.append(s)

    def switch_wait(self):
        """before nextw (requested 16 steps earlier) is consumed at a word switch.  vmcnt(0), not a count of the touches
        issued behind it: with two touches per step up to 66 vector loads of a wave are outstanding between two switches,
        the wave's VM counter holds 63, and a counted wait then let a stale genotype word through about once in a
        hundred launches (both blocks of one group wrong; found by tools/exp/repeat_determinism.py).  Waiting for
        everything every 16 steps costs 0.5 % and also bounds what is outstanding to 34."""
        self.e("s_waitcnt vmcnt(0)")

    def step(self, parity, windows, prefetch=True, pf=True):
        cur, nxt, sc, scn = regs_of(parity)
        e = self.e
        self.uid += 1
        uid = self.uid
        if prefetch:
            e(f"s_add_u32 s{S_DP}, s{S_DP}, %[stride]")
            e(f"s_addc_u32 s{S_DP + 1}, s{S_DP + 1}, 0")
            if "nosload" not in ABL and not ("halfsload" in ABL and parity == 1) and not ("quartersload" in ABL and self.uid % 4):
                e(f"s_load_dwordx16 {tup(nxt, 0)}, s[{S_DP}:{S_DP + 1}], 0x0")
                e(f"s_load_dwordx16 {tup(nxt, 1)}, s[{S_DP}:{S_DP + 1}], 0x40")
        if prefetch and self.gl:
            if parity == 0:
                self.gl_request()
            self.gl_read(scn, parity)
        elif prefetch:
            scnb = "scnb" if scn == "scn" else "scb"
            for blk, (vt, word, dst) in enumerate((("vt", "word", scn), ("vtb", "wordb", scnb))[:self.nb]):
                if "noint" not in ABL:
                    e(f"v_bfe_u32 %[{vt}], %[{word}], %[bit], 2")
                    e(f"v_lshl_add_u32 %[{vt}], %[{vt}], 3, %[row]")
                if "nolds" not in ABL:
                    e(f"ds_read_b64 %[{dst}], %[{vt}]")
            e("s_add_u32 %[row], %[row], 32")
            e("s_add_u32 %[bit], %[bit], 2")
            e("s_cmp_eq_u32 %[bit], 32")
            e(f"s_cbranch_scc0 WL_SAMEWORD_{uid}_%=")
            e("s_mov_b32 %[bit], 0")
            self.switch_wait()
            for word, nextw, gaddr in (("word", "nextw", "gaddr"), ("wordb", "nextwb", "gaddrb"))[:self.nb]:
                e(f"v_mov_b32_e32 %[{word}], %[{nextw}]")
                e(f"global_load_dword %[{nextw}], %[{gaddr}], off")
                e(f"v_lshl_add_u64 %[{gaddr}], %[{gaddr}], 0, %[rowbytes]")
            e(f"WL_SAMEWORD_{uid}_%=:")
            if PFW and self.nb == 2 and pf:
                e(f"s_cbranch_vccz WL_NOPF_{uid}_%=")
                for off in (0, 64, 124)[:PFW_LOADS]:      # [vz] = PFW * stride: the weights PFW steps ahead of s[S_DP]
                    e(f"global_load_dword %[vd], %[vz], s[{S_DP}:{S_DP + 1}] offset:{off}")
                e(f"WL_NOPF_{uid}_%=:")
        ws = list(windows)
        if self.nb == 2:
            # every weight multiplies the two blocks' scores: half the scalar loads per FP64 operation (the
            # scalar data path returns one dword per cycle and CU: at 16 weights per 32 operations it, not
            # the FP64 pipe, set the pace -- 0.61 of the FP64 peak, 0.78 with the loads halved)
            scb = "scb" if sc == "sc" else "scnb"
            for r in ws:
                e(f"v_mul_f64 %[t0], %[{sc}], {weight(cur, r)}")
                e(f"v_mul_f64 %[t1], %[{scb}], {weight(cur, r)}")
                e(f"v_add_f64 %[a{r}], %[a{r}], %[t0]")
                e(f"v_add_f64 %[b{r}], %[b{r}], %[t1]")
            ws = []
        for k in range(0, len(ws), 2):     # two products in flight: no back-to-back dependency
            pair = ws[k:k + 2]
            for t, r in zip(("t0", "t1"), pair):
                e(f"v_mul_f64 %[{t}], %[{sc}], {weight(cur, r)}")
            for t, r in zip(("t0", "t1"), pair):
                e(f"v_add_f64 %[a{r}], %[a{r}], %[{t}]")
        if prefetch and "nowait" not in ABL:
            e("s_waitcnt lgkmcnt(0)")


# Strip form of the GL variant (wlod_strip_gl_kernel): the compute waves of a workgroup walk ONE strip of
# windows of two 64-individual blocks in step -- wave k the 16-window groups k, k+N, .. -- and read the blocks'
# term rows from two rings in LDS that a loader wave fills once (LDS-DMA), so a row enters the CU once per strip
# instead of (W+15)/16 times.  Two blocks per wave as in the plain two-block loop.  Per step, instead of the
# genotype look-ups:  every other step publish the next row this wave reads ([vneed]);  if that row has not
# landed ([rown] >= [landed], the wave's copy of the loader's counter) poll the counter ([vflag]), at most
# [polls] times;  two ds_read_b64 (ring A at [lane8b] + [rd], ring B GLS_RING * 512 bytes above).
GLS_RING = int(os.environ.get("GARLIC_WLOD_GLS_RING", "32"))     # rows per ring
GLS_PFW = int(os.environ.get("GARLIC_WLOD_GLS_PFW", "1"))        # weight touches in the strip loop (2 more VGPRs)
GLS_MASK = GLS_RING * 512 - 1


class StripGen(Gen):
    def __init__(self, fast=False):
        super().__init__(gl=False, nb=2)
        self.fast = fast      # the 80-VGPR form: registers by number (FAST_REGS), one product register, write-out inside

    def read_next(self, dst, dstb, uid):
        e = self.e
        e("s_cmp_lt_u32 %[rown], %[landed]")
        e(f"s_cbranch_scc1 WS_OK_{uid}_%=")
        if "nopoll" in ABL:
            e(f"s_branch WS_OK_{uid}_%=")
        e(f"WS_POLL_{uid}_%=:")
        e(f"ds_read_b32 %[vtmp], %[lane8b] offset:{2 * GLS_RING * 512}")    # the lane's copy of the loader's counter
        e("s_waitcnt lgkmcnt(0)")
        e("v_readfirstlane_b32 %[landed], %[vtmp]")
        e("s_cmp_lt_u32 %[rown], %[landed]")
        e(f"s_cbranch_scc1 WS_OK_{uid}_%=")
        e("s_sub_u32 %[polls], %[polls], 1")
        e("s_cmp_eq_u32 %[polls], 0")
        e(f"s_cbranch_scc1 WS_OK_{uid}_%=")          # budget spent: go on (the caller traps)
        e("s_sleep 1")
        e(f"s_branch WS_POLL_{uid}_%=")
        e(f"WS_OK_{uid}_%=:")
        e("v_add_u32_e32 %[vt], %[rd], %[lane8b]")
        if "nolds" not in ABL:
            e(f"ds_read_b64 %[{dst}], %[vt]")
            e(f"ds_read_b64 %[{dstb}], %[vt] offset:{GLS_RING * 512}")
        e("s_add_u32 %[rd], %[rd], 512")
        e(f"s_and_b32 %[rd], %[rd], {GLS_MASK}")
        e("s_add_u32 %[rown], %[rown], 1")

    def step(self, parity, windows, prefetch=True, pf=True):
        cur, nxt, sc, scn = regs_of(parity)
        e = self.e
        self.uid += 1
        uid = self.uid
        if prefetch:
            e(f"s_add_u32 s{S_DP}, s{S_DP}, %[stride]")
            e(f"s_addc_u32 s{S_DP + 1}, s{S_DP + 1}, 0")
            if "nosload" not in ABL:
                e(f"s_load_dwordx16 {tup(nxt, 0)}, s[{S_DP}:{S_DP + 1}], 0x0")
                e(f"s_load_dwordx16 {tup(nxt, 1)}, s[{S_DP}:{S_DP + 1}], 0x40")
            if parity == 0:
                e("s_min_u32 %[stmp], %[rown], %[nextrow]")      # the wave's next group may start below this one's end
                e("v_add_u32_e32 %[vt], %[needoff], %[lane8b]")  # the lane's copy of this wave's row of need[]
                e("v_mov_b32_e32 %[vtmp], %[stmp]")
                e("ds_write_b32 %[vt], %[vtmp]")
            self.read_next(scn, "scnb" if scn == "scn" else "scb", uid)
            if GLS_PFW and pf and not self.fast:
                e(f"s_cbranch_vccz WL_NOPF_{uid}_%=")
                for off in (0, 64, 124)[:PFW_LOADS]:
                    e(f"global_load_dword %[vd], %[vz], s[{S_DP}:{S_DP + 1}] offset:{off}")
                # a throttle, not a wait for data (the touches are never read): at most 32 + this step's touches are ever
                # outstanding.  The VM counter holds 63: a counted wait past that lets requests through unseen (round 3's
                # 1-in-100 wrong group in the two-block loop)
                assert 32 + PFW_LOADS <= 63
                e("s_waitcnt vmcnt(32)")
                e(f"WL_NOPF_{uid}_%=:")
        scb = "scb" if sc == "sc" else "scnb"
        if "halffp" in ABL:
            windows = list(windows)[::2]
        for r in windows:
            if self.fast:
                # one product register: a dependent FP64 pair issues as fast as an independent one on gfx950
                # (profiles/r04_chain_body_ubench.txt: 4.3 cycles per v_add_f64 either way)
                e(f"v_mul_f64 %[t0], %[{sc}], {weight(cur, r)}")
                e(f"v_add_f64 %[a{r}], %[a{r}], %[t0]")
                e(f"v_mul_f64 %[t0], %[{scb}], {weight(cur, r)}")
                e(f"v_add_f64 %[b{r}], %[b{r}], %[t0]")
                continue
            e(f"v_mul_f64 %[t0], %[{sc}], {weight(cur, r)}")
            e(f"v_mul_f64 %[t1], %[{scb}], {weight(cur, r)}")
            e(f"v_add_f64 %[a{r}], %[a{r}], %[t0]")
            e(f"v_add_f64 %[b{r}], %[b{r}], %[t1]")
        if prefetch:
            e("s_waitcnt lgkmcnt(0)")


# ---- the 80-VGPR strip form (three workgroups per CU): every vector register by number, nothing comes back in registers.
# hipcc cannot keep 64 accumulators + the loop's temporaries as asm operands at 80 VGPRs (it spills every accumulator right
# behind the block, whatever the write-out looks like), so the block owns v4 .. v78 outright (clobbers), computes its lane
# addresses itself and ends with the write-out: per block, quarter by quarter, 16 lanes put their 16 sums into the wave's
# own patch in LDS (16 rows of WQ_PITCH bytes: no lock), then all 64 lanes take 16 B each -- 8 lanes one individual's
# 128 B -- and store them non-temporally.  v0 .. v3 and v79 stay with the compiler.
FAST_V_A, FAST_V_B = 4, 36
FAST_REGS = {"sc": "v[68:69]", "scb": "v[70:71]", "scn": "v[72:73]", "scnb": "v[74:75]", "t0": "v[76:77]",
             "vt": "v76", "vtmp": "v77",          # addresses / flag values of a step: dead before its first product
             "lane8b": "v78"}
for _r in range(R):
    FAST_REGS[f"a{_r}"] = f"v[{FAST_V_A + 2 * _r}:{FAST_V_A + 2 * _r + 1}]"
    FAST_REGS[f"b{_r}"] = f"v[{FAST_V_B + 2 * _r}:{FAST_V_B + 2 * _r + 1}]"
FAST_CLOBBER_V = list(range(4, 79))
WQ_PITCH = 144                 # bytes per patch row: 16 doubles + 16 B (wlod_strip_kernel.hpp: WT_PITCH * 8)
MISSING_HI = 0xC0C38780        # -9999.0


def fast_epilogue(e):
    Q, PW, PR, DST, ROWS, COLS, T = "v[68:71]", "v72", "v73", "v[74:75]", "v76", "v77", "v78"
    S_EXEC, S_C2, S_C1, S_ROW8, S_M2, S_M1 = "s[36:37]", "s[38:39]", "s[40:41]", "s[42:43]", "s[44:45]", "s[46:47]"
    # the next row this wave reads (its next group's first), published before anything else
    e("v_add_u32_e32 v76, %[needoff], v78")
    e("v_mov_b32_e32 v77, %[nextrow]")
    e("ds_write_b32 v76, v77")
    e(f"s_mov_b64 {S_EXEC}, exec")
    # MISSING where no scored window starts
    e("s_cmp_eq_u32 %[gm], 0xffff")
    e("s_cbranch_scc1 WF_SCORED_%=")
    for r in range(R):
        e(f"s_bitcmp1_b32 %[gm], {r}")
        e(f"s_cbranch_scc1 WF_KEEP{r}_%=")
        for base in (FAST_V_A, FAST_V_B):
            e(f"v_mov_b32_e32 v{base + 2 * r}, 0")
            e(f"v_mov_b32_e32 v{base + 2 * r + 1}, 0x{MISSING_HI:x}")
        e(f"WF_KEEP{r}_%=:")
    e("WF_SCORED_%=:")
    # lane addresses: patch row to write (lane & 15), patch piece to read (row lane >> 3, columns 2 (lane & 7) ..),
    # the piece's place in the score matrix, how many rows / columns exist from there
    e("v_mbcnt_lo_u32_b32 v77, -1, 0")
    e("v_mbcnt_hi_u32_b32 v77, -1, v77")
    e(f"v_lshrrev_b32_e32 {T}, 3, v77")                       # lane >> 3 (the ring address is dead)
    e("v_and_b32_e32 v76, 15, v77")
    e(f"v_mul_u32_u24_e32 {PW}, {WQ_PITCH}, v76")
    e(f"v_add_u32_e32 {PW}, %[wpatch], {PW}")
    e("v_and_b32_e32 v77, 7, v77")                            # lane & 7
    e(f"v_mul_u32_u24_e32 {PR}, {WQ_PITCH}, {T}")
    e(f"v_lshl_add_u32 {PR}, v77, 4, {PR}")
    e(f"v_add_u32_e32 {PR}, %[wpatch], {PR}")
    e(f"v_mad_u64_u32 {DST}, vcc, {T}, %[pitchb], 0")
    e("v_lshlrev_b32_e32 v76, 4, v77")
    e("v_add_co_u32_e32 v74, vcc, v74, v76")
    e("v_addc_co_u32_e32 v75, vcc, 0, v75, vcc")
    e(f"v_lshl_add_u64 {DST}, {DST}, 0, %[dst]")
    e(f"v_sub_u32_e32 {ROWS}, %[rows], {T}")                  # row 8 q + (lane >> 3) of the block exists iff 8 q < this
    e("v_lshlrev_b32_e32 v77, 1, v77")
    e(f"v_sub_u32_e32 {COLS}, %[cols], v77")                  # >= 2: both doubles of the piece exist; 1: the first
    e(f"v_cmp_lt_i32_e64 {S_C2}, 1, {COLS}")
    e(f"v_cmp_eq_u32_e64 {S_C1}, 1, {COLS}")
    e("s_lshl_b32 s42, %[pitchb], 3")                         # eight rows further down, bytes
    e("s_lshr_b32 s43, %[pitchb], 29")
    for blk, base in enumerate((FAST_V_A, FAST_V_B)):
        if blk == 1:
            e(f"v_subrev_u32_e32 {ROWS}, 64, {ROWS}")         # the second block's rows
        for h in range(4):
            lo, hi = ((0xffff << (16 * h)) & 0xffffffff, 0) if h < 2 else (0, (0xffff << (16 * (h - 2))) & 0xffffffff)
            e(f"s_mov_b32 exec_lo, 0x{lo:x}")
            e(f"s_mov_b32 exec_hi, 0x{hi:x}")
            for r in range(0, R, 2):
                e(f"ds_write2_b64 {PW}, v[{base + 2 * r}:{base + 2 * r + 1}], v[{base + 2 * r + 2}:{base + 2 * r + 3}] "
                  f"offset0:{r} offset1:{r + 1}")
            e(f"s_mov_b64 exec, {S_EXEC}")
            for j in range(2):
                k = 16 * h + 8 * j
                e(f"ds_read_b128 {Q}, {PR}" + (f" offset:{8 * WQ_PITCH}" if j else ""))
                e(f"v_cmp_lt_i32_e32 vcc, {k}, {ROWS}")
                e(f"s_and_b64 {S_M2}, vcc, {S_C2}")
                e(f"s_and_b64 {S_M1}, vcc, {S_C1}")
                e("s_waitcnt lgkmcnt(0)")
                e(f"s_mov_b64 exec, {S_M2}")
                e(f"global_store_dwordx4 {DST}, {Q}, off nt")
                e(f"s_cmp_lg_u64 {S_M1}, 0")
                e(f"s_cbranch_scc0 WF_NOODD{blk}_{h}_{j}_%=")
                e(f"s_mov_b64 exec, {S_M1}")                  # the chromosome's last window has an even index
                e(f"global_store_dwordx2 {DST}, v[68:69], off")
                e(f"WF_NOODD{blk}_{h}_{j}_%=:")
                e(f"s_mov_b64 exec, {S_EXEC}")
                e(f"v_lshl_add_u64 {DST}, {DST}, 0, {S_ROW8}")


def build_strip(fast=False):
    g = StripGen(fast)
    e = g.e
    if fast:
        e("v_mbcnt_lo_u32_b32 v78, -1, 0")
        e("v_mbcnt_hi_u32_b32 v78, -1, v78")
        e("v_lshl_add_u32 v78, v78, 3, %[ringlds]")         # the lane's place in a ring row
    e(f"s_mov_b64 s[{S_DP}:{S_DP + 1}], %[dp]")
    e(f"s_load_dwordx16 {tup('A', 0)}, s[{S_DP}:{S_DP + 1}], 0x0")
    e(f"s_load_dwordx16 {tup('A', 1)}, s[{S_DP}:{S_DP + 1}], 0x40")
    g.read_next("sc", "scb", 0)                       # row 0 of the group
    if GLS_PFW and not fast:
        e("s_cmp_lg_u32 %[pfon], 0")
        e("s_cselect_b64 vcc, -1, 0")
    for r in range(R):
        e(f"v_mov_b64_e32 %[b{r}], 0")
    for r in range(R):
        e(f"v_mov_b64_e32 %[a{r}], 0")
    e("s_waitcnt lgkmcnt(0)")
    for i in range(R - 1):
        g.step(i % 2, range(0, i + 1))
    e("WL_LOOP_%=:")
    g.step(1, range(R))
    e("s_sub_u32 %[n], %[n], 1")
    e("s_cmp_eq_u32 %[n], 0")
    e("s_cbranch_scc1 WL_TAIL_EVEN_%=")
    g.step(0, range(R))
    e("s_sub_u32 %[n], %[n], 1")
    e("s_cmp_lg_u32 %[n], 0")
    e("s_cbranch_scc1 WL_LOOP_%=")
    for first, label in ((1, None), (0, "WL_TAIL_EVEN_%=")):
        if label:
            e(label + ":")
        for d in range(R - 1):
            g.step((first + d) % 2, range(d + 1, R), prefetch=(d < R - 2), pf=False)
        if first == 1:
            e("s_branch WL_DONE_%=")
    e("WL_DONE_%=:")
    if fast:
        if "noepi" in ABL:
            e("v_add_u32_e32 v76, %[needoff], v78")
            e("v_mov_b32_e32 v77, %[nextrow]")
            e("ds_write_b32 v76, v77")
            e("s_waitcnt lgkmcnt(0)")
        else:
            fast_epilogue(e)
        import re
        return [re.sub(r"%\[(\w+)\]", lambda m: FAST_REGS.get(m.group(1), m.group(0)), ln) for ln in g.out]
    if GLS_PFW:
        e("s_waitcnt vmcnt(0)")                        # the touches' results (never read) have landed
    return g.out


def build(gl, nb=1):
    g = Gen(gl, nb)
    e = g.e
    assert not (gl and nb != 1)
    # ---- pipeline fill: weights and score of step 0
    if gl:
        e(f"s_mov_b64 s[{S_DP}:{S_DP + 1}], %[dp]")
        e(f"s_load_dwordx16 {tup('A', 0)}, s[{S_DP}:{S_DP + 1}], 0x0")
        e(f"s_load_dwordx16 {tup('A', 1)}, s[{S_DP}:{S_DP + 1}], 0x40")
        for _ in range(GL_AHEAD // 2):                   # rows 0 .. GL_AHEAD-1
            g.gl_request()
        e(f"s_waitcnt vmcnt({GL_AHEAD // 2 - 1})")       # rows 0, 1 have landed
        e("v_add_u32_e32 %[vt], %[rd], %[lane8b]")
        e("ds_read_b64 %[sc], %[vt]")
        e("s_add_u32 %[rd], %[rd], 512")
        e(f"s_and_b32 %[rd], %[rd], {GL_MASK}")
    else:
        # (the lane's first two genotype words are requested here too, so that their latency overlaps
        # with the first weights')
        blocks = (("word", "nextw", "gaddr", "vt", "sc"), ("wordb", "nextwb", "gaddrb", "vtb", "scb"))[:nb]
        for word, nextw, gaddr, _, _ in blocks:
            e(f"global_load_dword %[{word}], %[{gaddr}], off")
            e(f"global_load_dword %[{nextw}], %[{gaddr}], off offset:256")
        e(f"s_mov_b64 s[{S_DP}:{S_DP + 1}], %[dp]")
        e(f"s_load_dwordx16 {tup('A', 0)}, s[{S_DP}:{S_DP + 1}], 0x0")
        e(f"s_load_dwordx16 {tup('A', 1)}, s[{S_DP}:{S_DP + 1}], 0x40")
        for _, _, gaddr, _, _ in blocks:
            e(f"v_lshl_add_u64 %[{gaddr}], %[{gaddr}], 0, %[rowbytes]")
            e(f"v_lshl_add_u64 %[{gaddr}], %[{gaddr}], 0, %[rowbytes]")
        e("s_waitcnt vmcnt(0)")
        for word, _, _, vt, sc in blocks:
            e(f"v_bfe_u32 %[{vt}], %[{word}], %[bit], 2")
            e(f"v_lshl_add_u32 %[{vt}], %[{vt}], 3, %[row]")
            e(f"ds_read_b64 %[{sc}], %[{vt}]")
        e("s_add_u32 %[row], %[row], 32")
        e("s_add_u32 %[bit], %[bit], 2")
        e("s_cmp_eq_u32 %[bit], 32")
        e("s_cbranch_scc0 WL_SAMEWORD_0_%=")
        e("s_mov_b32 %[bit], 0")
        g.switch_wait()
        for word, nextw, gaddr, _, _ in blocks:
            e(f"v_mov_b32_e32 %[{word}], %[{nextw}]")
            e(f"global_load_dword %[{nextw}], %[{gaddr}], off")
            e(f"v_lshl_add_u64 %[{gaddr}], %[{gaddr}], 0, %[rowbytes]")
        e("WL_SAMEWORD_0_%=:")
    if nb == 2:
        if PFW:
            e("s_cmp_lg_u32 %[pfon], 0")      # VCC (not used otherwise) all ones: touch the weights ahead; 0: do not
            e("s_cselect_b64 vcc, -1, 0")
        for r in range(R):
            e(f"v_mov_b64_e32 %[b{r}], 0")
    for r in range(R):
        e(f"v_mov_b64_e32 %[a{r}], 0")
    e("s_waitcnt lgkmcnt(0)")
    # ---- windows enter one by one: steps 0..14
    for i in range(R - 1):
        g.step(i % 2, range(0, i + 1))
    # ---- steps 15 .. W-1: all windows (n = W - 15 >= 1 of them); step 15 is odd
    e("WL_LOOP_%=:")
    g.step(1, range(R))
    e("s_sub_u32 %[n], %[n], 1")
    e("s_cmp_eq_u32 %[n], 0")
    e("s_cbranch_scc1 WL_TAIL_EVEN_%=")          # next step index is even
    g.step(0, range(R))
    e("s_sub_u32 %[n], %[n], 1")
    e("s_cmp_lg_u32 %[n], 0")
    e("s_cbranch_scc1 WL_LOOP_%=")
    # ---- windows leave one by one: steps W .. W+14, first step odd (fall through) or even
    for first, label in ((1, None), (0, "WL_TAIL_EVEN_%=")):
        if label:
            e(label + ":")
        for d in range(R - 1):
            g.step((first + d) % 2, range(d + 1, R), prefetch=(d < R - 2), pf=False)
        if first == 1:
            e("s_branch WL_DONE_%=")
    e("WL_DONE_%=:")
    # whatever was requested ahead and never consumed (the look-ahead genotype word; the last term
    # rows) lands before the compiler reuses its register / the ring
    e("s_waitcnt vmcnt(0)")
    return g.out


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    # GARLIC_GEN_OUT: write somewhere else (tests/test_generated_cpu.py checks the committed file is current)
    path = os.path.join(os.environ.get("GARLIC_GEN_OUT") or os.path.join(here, "..", "garlic_amd", "csrc"), "wlod_loop_gfx950.inc")
    total = 0
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_wlod_asm.py -- do not edit; see that file for the schedule.\n")
        f.write("// wlod_group<16>: the ordered sums of 16 consecutive windows x 64 individuals, one SNP\n")
        f.write("// per step, the next step's weights and score requested before this step's FP64 work.\n")
        f.write("// _GL: the score is the lane's entry of the scaled TGLS term matrix, through an LDS ring.\n")
        f.write("// WLOD2: two 64-individual blocks per wave, every weight used for both (half the scalar loads per operation).\n")
        f.write(f"#define GARLIC_WLOD_GL_RING_ROWS {GL_RING}\n")
        f.write(f"#define GARLIC_WLOD_PFW {PFW}\n#define GARLIC_WLOD_PFW_MAX_W {PFW_MAX_W}\n")
        f.write(f"#define GARLIC_WLOD_GLS_RING_ROWS {GLS_RING}\n#define GARLIC_WLOD_GLS_PFW {GLS_PFW}\n")
        for name, gl, nb in (("GARLIC_WLOD_LOOP_ASM", False, 1), ("GARLIC_WLOD_GL_LOOP_ASM", True, 1),
                             ("GARLIC_WLOD2_LOOP_ASM", False, 2), ("GARLIC_WLOD_GLS_LOOP_ASM", None, 2),
                             ("GARLIC_WLOD_GLF_LOOP_ASM", "fast", 2)):
            lines = build_strip(gl == "fast") if gl in (None, "fast") else build(gl, nb)
            total += sum(1 for x in lines if not x.endswith(":"))
            f.write(f"#define {name} \\\n")
            for ln in lines:
                f.write('    "%s\\n\\t" \\\n' % ln)
            f.write('    ""\n')
        regs = ['"s%d"' % r for r in range(S_DP, 100)] + ['"scc"', '"vcc"']
        f.write("#define GARLIC_WLOD_LOOP_CLOBBERS \\\n    ")
        f.write(", \\\n    ".join(", ".join(regs[i:i + 12]) for i in range(0, len(regs), 12)) + "\n")
        vregs = ['"v%d"' % r for r in FAST_CLOBBER_V]
        f.write("#define GARLIC_WLOD_GLF_CLOBBERS GARLIC_WLOD_LOOP_CLOBBERS, \\\n    ")
        f.write(", \\\n    ".join(", ".join(vregs[i:i + 12]) for i in range(0, len(vregs), 12)) + "\n")
        f.write(f"#define GARLIC_WLOD_GLF_PATCH_PITCH_BYTES {WQ_PITCH}\n")
    print(f"wrote {os.path.normpath(path)}: {total} instructions in five variants")


if __name__ == "__main__":
    main()
