// wLOD with per-genotype likelihoods (src/garlic-roh.cpp:204-277 with USE_GL, :245), strip form.
//
// wlod_tile_glring_kernel gives every wave its own 16 windows x 64 individuals and its own copy of the term
// rows they need: a 512-B row of the term matrix enters a CU (W+15)/16 times (57 B per window at W = 100), the
// kernel's time is what HBM needs for the re-reads that miss L2 (115 of 147 GB at 2M x 1280) plus what the
// scalar data path needs for one block's weights.  Here a workgroup owns ONE strip of consecutive windows of a
// chromosome for TWO 64-individual blocks:
//
//   * N = 7 compute waves; wave k takes the 16-window groups k, k+N, k+2N, .. of the strip.  A group starting at
//     row 16 g (rows = SNPs, strip-relative) reads rows 16 g .. 16 g + W + 14, a wave's next group starts at row
//     16 (g + N): with W + 15 <= 16 N + 16 it starts at most 16 rows below the row its last one ended at, and all
//     waves move forward through the strip's rows together, within the rings' reach of each other.  (N + the
//     loader = 8 waves: two workgroups fit a CU.  With 8 + 1 waves only one did -- 96 VGPRs allow 5 waves per SIMD
//     and both workgroups put three on the same one --, and a loader wave that is not needed is not free either:
//     2M x 1280, W = 100: 9 waves 23.4 ms, 8 waves 21.7; narrower windows leave a wave idle between its groups
//     and are still faster with 7 waves than with ceil((W + 15) / 16): W = 50 13.0 against 14.0 ms.)
//   * one loader wave streams the two blocks' rows ONCE (LDS-DMA, 1 KB = 2 rows per request and block) into
//     two rings of WS_RING rows in LDS, never further ahead than the slowest compute wave allows
//     (need[k] = next row wave k reads, published every other step), and publishes how many rows have landed;
//   * a compute wave's loop is the two-block loop of wlod_tile2_kernel (every scalar-loaded weight multiplies
//     both blocks' scores; tools/gen_wlod_asm.py, GARLIC_WLOD_GLS_LOOP_ASM) with the scores read from the rings.
//
//   * scores into 16-B aligned rows at W <= 113 take wlod_strip_gl3_kernel (end of this file): the same strip at 80
//     VGPRs, THREE workgroups = 21 compute waves per CU.  hipcc spills every accumulator of a 64-accumulator asm
//     block at that budget, so the block (GARLIC_WLOD_GLF_LOOP_ASM) owns v4 .. v78 by number and writes the scores
//     out itself, through a patch per wave -- no lock; the loader touches the weights for everybody.
//     2M x 1280: W = 50 12.9 -> 12.0 ms, W = 100 21.9 -> 21.1-21.4, W = 113 -> 23.3.  What it does not buy is clock: the
//     chip sits on its power cap either way (two per CU 2.11-2.14 GHz, three 1.92 GHz at 1283 W,
//     profiles/r04_wlodgl_ablations_clock.txt) -- at its clock the loop is at 0.76 of the FP64 rate.
//
// HBM traffic: 8 B of terms per window and individual, once (+ W / strip length), instead of 57 B.
// Arithmetic and write-out are those of the tile kernels: bit-identical output.
//
// Progress: the compute wave with the smallest need[] only ever waits for rows need, need + 1; the loader may
// always issue them (3 <= WS_RING), and while it has no room it keeps retiring and publishing what it has in
// flight.  Waits are bounded all the same (a spent budget traps instead of hanging the GPU).
#pragma once
#include "tgls_ring_kernel.hpp"
#include "variant_kernels.hpp"

namespace garlic {

constexpr int WS_WAVES = 7;                             // compute waves per workgroup (+ the loader: 8 waves, two workgroups per CU)
constexpr int WS_WAVES_WIDE = 15;                       // ... for 113 < W <= 241: 16 waves, one workgroup per CU (the same 14-15 compute waves per CU)
constexpr int WS_NEED_ROWS = 16;
constexpr int WS_RING = GARLIC_WLOD_GLS_RING_ROWS;      // rows per ring (power of two)
constexpr int WS_DEPTH = 6;                             // loader: row pairs (per block) in flight
constexpr int WS_NEVER = 0x7fffffff;
constexpr uint32_t WS_RING_BYTES = (uint32_t)WS_RING * WAVE * 8u;
// dynamic LDS: ring A, ring B (1-KB aligned), the loader's counter (one 512-B row: every lane reads its own copy,
// at its ring address + an immediate offset -- no address register), need[16] (sixteen such rows), patch lock
// (16 B), patch [64][WT_PITCH]
constexpr uint32_t WS_LANDED_OFF = 2u * WS_RING_BYTES;
constexpr uint32_t WS_NEED_OFF = WS_LANDED_OFF + 512u;
constexpr uint32_t WS_LOCK_OFF = WS_NEED_OFF + (uint32_t)WS_NEED_ROWS * 512u;
constexpr uint32_t WS_PATCH_OFF = WS_LOCK_OFF + 16u;
constexpr uint32_t WS_LDS_BYTES = WS_PATCH_OFF + (uint32_t)(WAVE * WT_PITCH * 8);
// the 80-VGPR form (wlod_strip_gl3_kernel: three workgroups per CU): need[8], no lock, a patch of 16 rows per compute wave
constexpr int WF_NEED_ROWS = 8;
constexpr uint32_t WF_PATCH_BYTES = 16u * GARLIC_WLOD_GLF_PATCH_PITCH_BYTES;
constexpr uint32_t WF_PATCH_OFF = WS_NEED_OFF + (uint32_t)WF_NEED_ROWS * 512u;
constexpr uint32_t WF_LDS_BYTES = WF_PATCH_OFF + (uint32_t)WS_WAVES * WF_PATCH_BYTES;
static_assert(GARLIC_WLOD_GLF_PATCH_PITCH_BYTES == WT_PITCH * 8, "the generated write-out and the patch agree on the pitch");
static_assert(3u * ((WF_LDS_BYTES + 1279u) / 1280u * 1280u) <= 160u * 1024u, "three workgroups' LDS fit a CU");

struct WlodStrip {
    int32_t chr, s_begin, n_groups, pad;               // windows s_begin .. s_begin + 16 * n_groups - 1 of the chromosome
};

struct WlodStripArgs {
    const uint8_t *valid;      // per SNP: a scored window starts here
    const ChrDev *chrs;
    const WlodStrip *strips;
    const double *terms;       // scaled term matrix [blk][term_rows][64]
    const double *D;           // skewed reciprocal weights (+ SKEW_FRONT)
    double *out;
    int64_t term_rows;
    int32_t ind_begin, ind_count, winsize, n_waves, n_pairs, use_patch;
    uint32_t n_work;           // strips x pairs
    int32_t *stalled;          // set to 1 when a wave gives up waiting (its results are then wrong): the host reruns the
                               // call with the tile form.  A poll budget measures time, not progress -- under a
                               // counter-serialising profiler a correct run may exhaust it -- so nothing traps.
    CovBits cov;               // coverage bits instead of scores (variant_kernels.hpp); bits == NULL: scores
};

// flag rows: lane i's copy at row + 8 i
__device__ __forceinline__ void ws_row_write(uint32_t lane8b, uint32_t off, int v)
{
    asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(lane8b + off), "v"(v) : "memory");
}
template <int ROWS>
__device__ __forceinline__ int ws_min_need(uint32_t lane8b);
template <>
__device__ __forceinline__ int ws_min_need<8>(uint32_t lane8b)
{
    int x[8];
    const uint32_t a = lane8b + WS_NEED_OFF;
    asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:512\n\tds_read_b32 %2, %8 offset:1024\n\t"
                 "ds_read_b32 %3, %8 offset:1536\n\tds_read_b32 %4, %8 offset:2048\n\tds_read_b32 %5, %8 offset:2560\n\t"
                 "ds_read_b32 %6, %8 offset:3072\n\tds_read_b32 %7, %8 offset:3584\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7])
                 : "v"(a) : "memory");
    int m = x[0];
#pragma unroll
    for (int k = 1; k < 8; k++) m = min(m, x[k]);
    return __builtin_amdgcn_readfirstlane(m);
}
template <>
__device__ __forceinline__ int ws_min_need<16>(uint32_t lane8b)
{   // all sixteen rows (rows of waves the workgroup does not have hold WS_NEVER), one wait
    static_assert(WS_NEED_ROWS == 16, "sixteen rows of need[] are read");
    int x[16];
    const uint32_t a = lane8b + WS_NEED_OFF;
    asm volatile("ds_read_b32 %0, %16\n\tds_read_b32 %1, %16 offset:512\n\tds_read_b32 %2, %16 offset:1024\n\t"
                 "ds_read_b32 %3, %16 offset:1536\n\tds_read_b32 %4, %16 offset:2048\n\tds_read_b32 %5, %16 offset:2560\n\t"
                 "ds_read_b32 %6, %16 offset:3072\n\tds_read_b32 %7, %16 offset:3584\n\tds_read_b32 %8, %16 offset:4096\n\t"
                 "ds_read_b32 %9, %16 offset:4608\n\tds_read_b32 %10, %16 offset:5120\n\tds_read_b32 %11, %16 offset:5632\n\t"
                 "ds_read_b32 %12, %16 offset:6144\n\tds_read_b32 %13, %16 offset:6656\n\tds_read_b32 %14, %16 offset:7168\n\t"
                 "ds_read_b32 %15, %16 offset:7680\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]),
                   "=&v"(x[8]), "=&v"(x[9]), "=&v"(x[10]), "=&v"(x[11]), "=&v"(x[12]), "=&v"(x[13]), "=&v"(x[14]), "=&v"(x[15])
                 : "v"(a) : "memory");
    int m = x[0];
#pragma unroll
    for (int k = 1; k < 16; k++) m = min(m, x[k]);
    return __builtin_amdgcn_readfirstlane(m);
}

// the two-block loop over one 16-window group, scores from the rings (see tools/gen_wlod_asm.py)
__device__ __forceinline__ void wlod_group_gls(uint32_t lane8b, uint32_t needoff, const double *Ds, int W,
                                               uint32_t row0, uint32_t nextrow, uint32_t &landed, uint32_t &polls,
                                               double (&acc)[WLOD_R], double (&bcc)[WLOD_R])
{
    constexpr int R = WLOD_R;
    static_assert(R == 16, "the hand-scheduled loop keeps 16 weights per step in SGPRs");
    double sc, scn, scb, scnb, t0, t1;
    uint32_t vt, vtmp, stmp;
#if GARLIC_WLOD_GLS_PFW
    uint32_t vd;
#endif
    uint32_t n = (uint32_t)(W - (R - 1));
    uint32_t rown = row0, rd = (row0 & (uint32_t)(WS_RING - 1)) * 512u;
    const double *dp = Ds - (R - 1);
    const uint32_t stride = (uint32_t)(W + 1) * 8u;
    asm volatile(GARLIC_WLOD_GLS_LOOP_ASM
                 : [a0] "=&v"(acc[0]), [a1] "=&v"(acc[1]), [a2] "=&v"(acc[2]), [a3] "=&v"(acc[3]),
                   [a4] "=&v"(acc[4]), [a5] "=&v"(acc[5]), [a6] "=&v"(acc[6]), [a7] "=&v"(acc[7]),
                   [a8] "=&v"(acc[8]), [a9] "=&v"(acc[9]), [a10] "=&v"(acc[10]), [a11] "=&v"(acc[11]),
                   [a12] "=&v"(acc[12]), [a13] "=&v"(acc[13]), [a14] "=&v"(acc[14]), [a15] "=&v"(acc[15]),
                   [b0] "=&v"(bcc[0]), [b1] "=&v"(bcc[1]), [b2] "=&v"(bcc[2]), [b3] "=&v"(bcc[3]),
                   [b4] "=&v"(bcc[4]), [b5] "=&v"(bcc[5]), [b6] "=&v"(bcc[6]), [b7] "=&v"(bcc[7]),
                   [b8] "=&v"(bcc[8]), [b9] "=&v"(bcc[9]), [b10] "=&v"(bcc[10]), [b11] "=&v"(bcc[11]),
                   [b12] "=&v"(bcc[12]), [b13] "=&v"(bcc[13]), [b14] "=&v"(bcc[14]), [b15] "=&v"(bcc[15]),
                   [sc] "=&v"(sc), [scn] "=&v"(scn), [scb] "=&v"(scb), [scnb] "=&v"(scnb), [t0] "=&v"(t0), [t1] "=&v"(t1),
                   [vt] "=&v"(vt), [vtmp] "=&v"(vtmp), [rd] "+s"(rd), [rown] "+s"(rown),
                   [landed] "+s"(landed), [polls] "+s"(polls), [n] "+s"(n), [stmp] "=&s"(stmp)
#if GARLIC_WLOD_GLS_PFW
                   , [vd] "=&v"(vd)
#endif
                 : [nextrow] "s"(nextrow), [dp] "s"(dp), [stride] "s"(stride), [lane8b] "v"(lane8b), [needoff] "s"(needoff)
#if GARLIC_WLOD_GLS_PFW
                   , [vz] "v"((uint32_t)GARLIC_WLOD_PFW * stride),
                   [pfon] "s"(__builtin_amdgcn_readfirstlane(W <= GARLIC_WLOD_PFW_MAX_W ? 1 : 0))
#endif
                 : GARLIC_WLOD_LOOP_CLOBBERS, "memory");
}

// waits until at most `pairs` of the loader's row pairs (two requests each) are still in flight
// ... three requests each when the loader also touches the pairs' weights (all loads: they retire in issue order)
__device__ __forceinline__ void ws_wait_pairs3(int pairs)
{
    static_assert(3 * WS_DEPTH <= 63, "the VM counter holds 63");
    switch (pairs) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
    }
}

__device__ __forceinline__ void ws_wait_pairs(int pairs)
{
    static_assert(WS_DEPTH <= 8, "one s_waitcnt per possible count");
    switch (pairs) {          // requests retire in issue order; the count is an immediate
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    }
}

// loader wave: rows 0 .. n_rows-1 (n_rows even) of both blocks into the rings, two rows per request
// TOUCH: with every pair of rows, the weights the compute waves multiply them by -- rows r, r + 1 of the strip meet
// entries -15 .. W-1 of the weight rows at wrow + r W (tools/gen_wlod_asm.py: step i of the group at s reads 16 doubles at
// D + (s + i) W + i - 15) -- are touched, one dword per 128-B line, so that the scalar loads find them in L2.  (The
// other kernels' waves touch their own weights; here that would be seven waves touching the same lines, and two more
// registers in a loop that has none to spare.)
template <int NEED_ROWS, bool TOUCH>
__device__ __forceinline__ void ws_loader(const double *srcA, const double *srcB, int n_rows, uint32_t ring_lds, int lane,
                                          int32_t *stalled, const double *wrow = nullptr, int W = 0)
{
    bool gave_up = false;
    uint32_t touched = 0;                                   // lives in one register from the first touch to the last wait
    const int touch_lines = (2 * W * 8 + 120 + 127) / 128;   // (+ 120: the 15 entries before the row)
    const char *tp = reinterpret_cast<const char *>(wrow) - 120 + (int64_t)lane * 128;
    const uint32_t lane16 = (uint32_t)lane * 16u, lane8b = ring_lds + (uint32_t)lane * 8u;
    uint32_t slot_off = 0;
    int inflight = 0, published = 0, min_need = 0;
    for (int r = 0; r < n_rows; r += 2) {
        if (!gave_up && r + 2 - min_need > WS_RING) {
            // no room: meanwhile retire and publish what is in flight, oldest first (a compute wave may be waiting
            // for exactly those rows -- never block with unpublished rows)
            int budget = 1 << 22;
            while (r + 2 - (min_need = ws_min_need<NEED_ROWS>(lane8b)) > WS_RING) {
                if (inflight > 0) {
                    if (TOUCH) ws_wait_pairs3(--inflight); else ws_wait_pairs(--inflight);
                    published += 2;
                    ws_row_write(lane8b, WS_LANDED_OFF, published);
                } else {
                    if (--budget == 0) {     // say so and stream on without waiting: every wave drains, the host reruns
                        if (lane == 0) __hip_atomic_store(stalled, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        gave_up = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
        }
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     :: "s"(ring_lds + slot_off), "v"(lane16), "s"(srcA) : "memory");
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     :: "s"(ring_lds + WS_RING_BYTES + slot_off), "v"(lane16), "s"(srcB) : "memory");
        if (TOUCH) {
            // (a request by every lane or by none: the count of requests per pair is what the waits rely on)
            const char *t = lane < touch_lines ? tp : reinterpret_cast<const char *>(wrow);
            asm volatile("global_load_dword %0, %1, off" : "+v"(touched) : "v"(t) : "memory");
            tp += 2 * (int64_t)W * 8;
        }
        slot_off = (slot_off + 1024u) & (WS_RING_BYTES - 1u);
        srcA += 2 * WAVE;
        srcB += 2 * WAVE;
        if (++inflight == WS_DEPTH) {       // the oldest pair has landed
            if (TOUCH) ws_wait_pairs3(--inflight); else ws_wait_pairs(--inflight);
            published += 2;
            ws_row_write(lane8b, WS_LANDED_OFF, published);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(touched) :: "memory");
    ws_row_write(lane8b, WS_LANDED_OFF, n_rows);
}

// NW = 7: 8 waves, 5 waves per SIMD = at most 96 VGPRs (at 80 -- 6 waves, three workgroups per CU -- hipcc does not
// finish allocating registers around the loop); NW = 15: 16 waves, one workgroup per CU
template <bool ALIGNED16, int NW>
__global__ void __launch_bounds__((NW + 1) * WAVE, NW == WS_WAVES ? 5 : 4)
wlod_strip_gl_kernel(WlodStripArgs p)
{
    extern __shared__ __attribute__((aligned(1024))) char ws_lds[];
    const int lane = threadIdx.x & (WAVE - 1), W = p.winsize;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = p.n_waves;                              // blockDim.x = (N + 1) * 64
    int *patch_lock = reinterpret_cast<int *>(ws_lds + WS_LOCK_OFF);
    double *patch = reinterpret_cast<double *>(ws_lds + WS_PATCH_OFF);
    int *landed = reinterpret_cast<int *>(ws_lds + WS_LANDED_OFF);     // [64][2]: a copy per lane
    int *need = reinterpret_cast<int *>(ws_lds + WS_NEED_OFF);         // [16][64][2]
    // one contiguous range of the work per XCD (one L2 each): the pairs of a strip share its weights
    const unsigned per_xcd = gridDim.x >> 3;
    const unsigned v = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (v >= p.n_work) return;
    const WlodStrip st = p.strips[v / (unsigned)p.n_pairs];
    const int pair = (int)(v % (unsigned)p.n_pairs);
    const ChrDev c = p.chrs[st.chr];
    if (threadIdx.x == 0) *patch_lock = 0;
    if (wave == 0) landed[2 * lane] = 0;
    for (int r = wave; r < WS_NEED_ROWS; r += N + 1) need[(r * WAVE + lane) * 2] = (r < N && r < st.n_groups) ? 16 * r : WS_NEVER;
    __syncthreads();
    const int ind0A = pair * 2 * WAVE, ind0B = ind0A + WAVE;
    const bool activeB = ind0B < p.ind_count;
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)ws_lds;
    const int64_t blkA = ((int64_t)p.ind_begin + ind0A) >> 6;          // block-aligned shard (host-checked)
    const int64_t G0 = c.loc_base + GOFF + st.s_begin;                  // padded row of the strip's row 0
    if (wave == N) {
        const double *srcA = p.terms + (blkA * p.term_rows + G0) * WAVE;
        const double *srcB = activeB ? srcA + p.term_rows * WAVE : srcA;   // no second block: the first one again
        const int n_rows = (16 * (st.n_groups - 1) + W + 15 + 1) & ~1;
        __builtin_amdgcn_s_setprio(3);     // few instructions, and everybody waits for them
        ws_loader<WS_NEED_ROWS, false>(srcA, srcB, n_rows, ring_lds, lane, p.stalled);
        return;
    }
    const uint32_t lane8b = ring_lds + (uint32_t)lane * 8u;
    const uint32_t needoff = WS_NEED_OFF + (uint32_t)wave * 512u;
    uint32_t landed_seen = 0;
#pragma unroll 1
    for (int g = wave; g < st.n_groups; g += N) {
        const int s = st.s_begin + 16 * g;
        const bool has = lane < WLOD_R && s + lane < c.nloci && p.valid[c.loc_base + s + lane] != 0;
        const uint32_t gm = (uint32_t)__ballot(has) & 0xffffu;
        double acc[WLOD_R], bcc[WLOD_R];
        const int next_row = g + N < st.n_groups ? 16 * (g + N) : WS_NEVER;
        if (gm != 0) {
            uint32_t polls = 1u << 20;
            wlod_group_gls(lane8b, needoff, p.D + (c.loc_base + s) * (int64_t)W, W, (uint32_t)(16 * g),
                           (uint32_t)next_row, landed_seen, polls, acc, bcc);
            if (polls == 0) {            // rows that never came: flag the launch, release the loader, leave
                if (lane == 0) __hip_atomic_store(p.stalled, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ws_row_write(lane8b, needoff, WS_NEVER);
                break;
            }
        }
        ws_row_write(lane8b, needoff, next_row);
        wlod_write_group<WLOD_R, ALIGNED16>(acc, gm, c, p, p.out, patch, patch_lock, ind0A, s, 0, lane, st.chr);
        if (activeB) wlod_write_group<WLOD_R, ALIGNED16>(bcc, gm, c, p, p.out, patch, patch_lock, ind0B, s, 0, lane, st.chr);
    }
}

// ---- three workgroups per CU: 80 VGPRs, scores only, 16-B aligned rows, W <= 113 ---------------------------------------
// The same strip, rings, loader and flags.  What differs: the compute waves' loop owns its vector registers by number and
// ends with the write-out (tools/gen_wlod_asm.py, GARLIC_WLOD_GLF_LOOP_ASM), through a patch of the wave's own: no lock,
// no accumulator ever an operand hipcc has to place.  21 compute waves per CU instead of 14.
__device__ __forceinline__ void wlod_group_glf(uint32_t ring_lds, uint32_t needoff, const double *Ds, int W,
                                               uint32_t row0, uint32_t nextrow, uint32_t &landed, uint32_t &polls,
                                               uint32_t wpatch, double *dst, uint32_t pitch_bytes, int rows, int cols, uint32_t gm)
{
    constexpr int R = WLOD_R;
    static_assert(R == 16, "the hand-scheduled loop keeps 16 weights per step in SGPRs");
    uint32_t stmp;
    uint32_t n = (uint32_t)(W - (R - 1));
    uint32_t rown = row0, rd = (row0 & (uint32_t)(WS_RING - 1)) * 512u;
    const double *dp = Ds - (R - 1);
    const uint32_t stride = (uint32_t)(W + 1) * 8u;
    asm volatile(GARLIC_WLOD_GLF_LOOP_ASM
                 : [rd] "+s"(rd), [rown] "+s"(rown), [landed] "+s"(landed), [polls] "+s"(polls), [n] "+s"(n), [stmp] "=&s"(stmp)
                 : [nextrow] "s"(nextrow), [dp] "s"(dp), [stride] "s"(stride), [needoff] "s"(needoff), [ringlds] "s"(ring_lds),
                   [wpatch] "s"(wpatch), [dst] "s"(dst), [pitchb] "s"(pitch_bytes), [rows] "s"(rows), [cols] "s"(cols), [gm] "s"(gm)
                 : GARLIC_WLOD_GLF_CLOBBERS, "memory");
}

// a group in which no scored window starts: MISSING over the lane's own row piece
__device__ __forceinline__ void ws_write_missing(const ChrDev &c, const WlodStripArgs &p, int ind0, int s0)
{
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    if (ind0 + lane >= p.ind_count) return;
    double *out_row = p.out + c.out_base + (int64_t)(ind0 + lane) * c.out_pitch + s0;
#pragma unroll 1
    for (int r = 0; r < WLOD_R; r++)
        if (s0 + r < c.nloci) out_row[r] = MISSING_D;
}

__global__ void __launch_bounds__((WS_WAVES + 1) * WAVE, 6)    // 6 waves per SIMD: at most 80 VGPRs
wlod_strip_gl3_kernel(WlodStripArgs p)
{
    extern __shared__ __attribute__((aligned(1024))) char ws_lds[];
    constexpr int N = WS_WAVES;
    const int lane = threadIdx.x & (WAVE - 1), W = p.winsize;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int *landed = reinterpret_cast<int *>(ws_lds + WS_LANDED_OFF);
    int *need = reinterpret_cast<int *>(ws_lds + WS_NEED_OFF);         // [8][64][2]
    const unsigned per_xcd = gridDim.x >> 3;
    const unsigned v = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (v >= p.n_work) return;
    const WlodStrip st = p.strips[v / (unsigned)p.n_pairs];
    const int pair = (int)(v % (unsigned)p.n_pairs);
    const ChrDev c = p.chrs[st.chr];
    if (wave == 0) landed[2 * lane] = 0;
    need[(wave * WAVE + lane) * 2] = (wave < N && wave < st.n_groups) ? 16 * wave : WS_NEVER;
    __syncthreads();
    const int ind0A = pair * 2 * WAVE, ind0B = ind0A + WAVE;
    const bool activeB = ind0B < p.ind_count;
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)ws_lds;
    const int64_t blkA = ((int64_t)p.ind_begin + ind0A) >> 6;
    const int64_t G0 = c.loc_base + GOFF + st.s_begin;
    if (wave == N) {
        const double *srcA = p.terms + (blkA * p.term_rows + G0) * WAVE;
        const double *srcB = activeB ? srcA + p.term_rows * WAVE : srcA;
        const int n_rows = (16 * (st.n_groups - 1) + W + 15 + 1) & ~1;
        __builtin_amdgcn_s_setprio(3);
        ws_loader<WF_NEED_ROWS, true>(srcA, srcB, n_rows, ring_lds, lane, p.stalled,
                                      p.D + (c.loc_base + st.s_begin) * (int64_t)W, W);
        return;
    }
    const uint32_t needoff = WS_NEED_OFF + (uint32_t)wave * 512u;
    const uint32_t wpatch = ring_lds + WF_PATCH_OFF + (uint32_t)wave * WF_PATCH_BYTES;
    const uint32_t pitch_bytes = (uint32_t)(c.out_pitch * 8);          // host-checked: below 4 GB
    uint32_t landed_seen = 0;
#pragma unroll 1
    for (int g = wave; g < st.n_groups; g += N) {
        const int s = st.s_begin + 16 * g;
        const bool has = lane < WLOD_R && s + lane < c.nloci && p.valid[c.loc_base + s + lane] != 0;
        const uint32_t gm = (uint32_t)__ballot(has) & 0xffffu;
        const int next_row = g + N < st.n_groups ? 16 * (g + N) : WS_NEVER;
        if (gm == 0) {                   // nothing to sum
            ws_row_write(ring_lds + (uint32_t)lane * 8u, needoff, next_row);
            ws_write_missing(c, p, ind0A, s);
            if (activeB) ws_write_missing(c, p, ind0B, s);
            continue;
        }
        uint32_t polls = 1u << 20;
        wlod_group_glf(ring_lds, needoff, p.D + (c.loc_base + s) * (int64_t)W, W, (uint32_t)(16 * g), (uint32_t)next_row,
                       landed_seen, polls, wpatch, p.out + c.out_base + (int64_t)ind0A * c.out_pitch + s, pitch_bytes,
                       p.ind_count - ind0A, c.nloci - s, gm);
        if (polls == 0) {                // rows that never came: flag the launch, release the loader, leave
            if (lane == 0) __hip_atomic_store(p.stalled, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ws_row_write(ring_lds + (uint32_t)lane * 8u, needoff, WS_NEVER);
            break;
        }
    }
}

} // namespace garlic
