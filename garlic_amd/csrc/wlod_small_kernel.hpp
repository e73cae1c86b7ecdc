// wLOD (garlic-roh.cpp:204-277) for windows narrower than a 16-window group, 2 <= W <= 15 -- GARLIC's default
// --winsize is 10.  These shapes are bound by their output (8 B per window against 2 W flops), so the kernel is
// built like a copy: nothing inside its window loop waits for memory.
//
// wlod_tile_small_kernel (variant_kernels.hpp) ran at 0.36 of the HBM rate at 10M SNPs x 1250: per 32-window tile a
// staging pass and a barrier, per group ~W + 15 genotype look-ups behind global loads, the weights through scalar
// loads the compiler waits for where it issues them (a scalar load's latency can only be covered by ONE step of
// arithmetic -- they return out of order, so every wait is lgkmcnt(0) -- and a step here is 2 W instructions),
// and a write-out patch shared by the workgroup behind a lock: compute alone 4.6 ms, stores alone 4.0, together
// 7.5-8.7 at 2M x 1280.
//
// Here a workgroup owns WSM_T = 256 consecutive windows of a chromosome for eight 64-individual blocks (4 waves x
// 2 blocks).  Once per workgroup, coalesced: the score rows of its W + 255 SNPs (32 B each) and the plain
// reciprocal weights of its 256 windows (rld[s][0 .. W-1], contiguous: W * 8 B per window) go to LDS.  Then every
// wave walks the windows 16 at a time: the lane's score of a SNP is one LDS look-up by genotype (the genotype words
// of the next group are requested before this group's arithmetic), the weights of a group come from LDS 64 at a time, a lane each, and reach
// the multiplications through v_readlane (shared by the wave's two blocks), the sum runs j = 0 .. W-1 from +0.0 with the product rounded before the
// add (garlic-roh.cpp:262-268).  Write-out: a patch of the wave's own, 16 individuals at a time, 128 contiguous
// bytes per row and store, non-temporal (no lock, no barrier: a wave's LDS operations execute in order).
#pragma once
#include "variant_kernels.hpp"

namespace garlic {

constexpr int WSM_T = 256;                      // windows per workgroup
constexpr int WSM_ROWS = WSM_T + 16;            // staged score rows (W - 1 <= 14 beyond the last window; whole groups)
constexpr int WSM_PATCH_ROWS = 16;              // rows of a wave's write-out patch
constexpr int WSM_PATCH = WSM_PATCH_ROWS * WT_PITCH;

__host__ __device__ constexpr size_t wlod_small_lds_bytes(int W)
{
    return sizeof(double) * ((size_t)WSM_ROWS * 4 + (size_t)WSM_T * W + (size_t)WLOD_WAVES * WSM_PATCH);
}

// one group (16 windows x 64 individuals) out through the wave's patch; gm: bit r = window r holds a score
template <bool WHOLE>
__device__ __forceinline__ void wlod_small_write(const double (&acc)[16], uint32_t gm, const ChrDev &c, const WlodArgs &p,
                                                 double *__restrict__ out, double *mpatch, int ind0, int sg, int lane)
{
    const int sub = lane >> 4, cc = 2 * (lane & 7), r8 = lane >> 3;
    double *wr = mpatch + (lane & 15) * WT_PITCH;
    const double *rd = mpatch + r8 * WT_PITCH + cc;
    const uint64_t pitch = (uint64_t)(uint32_t)c.out_pitch;      // (a chromosome's pitch is below 2^31 elements)
    double *dst = out + c.out_base + (int64_t)ind0 * c.out_pitch + sg + cc + (uint64_t)(uint32_t)r8 * pitch;
#pragma unroll
    for (int pass = 0; pass < WAVE / WSM_PATCH_ROWS; pass++) {
        if (sub == pass) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) *reinterpret_cast<double2 *>(wr + r) = make_double2(acc[r], acc[r + 1]);
            if (gm != 0xFFFFu) {            // windows without a score: MISSING (garlic-roh.cpp:232); rare
#pragma unroll 1
                for (int r = 0; r < 16; r++)
                    if (!((gm >> r) & 1u)) wr[r] = MISSING_D;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const double2 v0 = *reinterpret_cast<const double2 *>(rd);
        const double2 v1 = *reinterpret_cast<const double2 *>(rd + 8 * WT_PITCH);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (WHOLE) {
            __builtin_nontemporal_store(v0.x, dst);
            __builtin_nontemporal_store(v0.y, dst + 1);
            __builtin_nontemporal_store(v1.x, dst + 8 * pitch);
            __builtin_nontemporal_store(v1.y, dst + 8 * pitch + 1);
        } else {
            const int row0 = ind0 + pass * WSM_PATCH_ROWS + r8;
            const bool c0 = sg + cc < c.nloci, c1 = sg + cc + 1 < c.nloci;
            if (row0 < p.ind_count) {
                if (c1) { __builtin_nontemporal_store(v0.x, dst); __builtin_nontemporal_store(v0.y, dst + 1); }
                else if (c0) dst[0] = v0.x;
            }
            if (row0 + 8 < p.ind_count) {
                if (c1) { __builtin_nontemporal_store(v1.x, dst + 8 * pitch); __builtin_nontemporal_store(v1.y, dst + 8 * pitch + 1); }
                else if (c0) dst[8 * pitch] = v1.x;
            }
        }
        dst += WSM_PATCH_ROWS * pitch;
    }
}

// the 16 + WC - 1 SNPs of a group as two 32-bit pieces of the lane's genotype stream, SNP i of the group at bits
// 2 i (i < 16: lo, else hi): three words funnel-shifted by the group's offset in its first word
struct WsmGeno { uint32_t lo, hi; };
__device__ __forceinline__ WsmGeno wsm_align(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t shift)
{
    WsmGeno g;
    g.lo = __builtin_amdgcn_alignbit(w1, w0, shift);      // ({w1, w0} >> shift)[31:0]; shift = 0 gives w0
    g.hi = __builtin_amdgcn_alignbit(w2, w1, shift);
    return g;
}

// GL: per-genotype likelihoods -- the lane's score of a SNP is its own entry of the scaled term matrix (wtab =
// [block][row][64], p.score_rows rows per block) instead of a look-up by genotype.  wlod_tile_small_gl_kernel read the
// W + 15 rows of every 16-window group afresh, 1.56 x the matrix at W = 10, and they came from memory every time (2M x
// 1280: FETCH x 2 = 32 GB for 20.5 GB of terms; the kernel moved 6.3 TB/s and still took 8.3 ms).  Here a wave walks 256
// consecutive windows and carries the W - 1 rows two neighbouring groups share in registers: 1.04 x.
template <int WC, bool GL>
__device__ __forceinline__ void
wlod_small_body(const uint32_t *__restrict__ packed, const double *__restrict__ wtab, const double *__restrict__ rld,
                double *__restrict__ out, const WlodArgs &p, double *dyn)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *rows = dyn;                                   // [WSM_ROWS][4]
    double *wts = dyn + WSM_ROWS * 4;                     // [WSM_T][WC]
    double *mpatch = wts + WSM_T * WC + wave * WSM_PATCH;
    const unsigned per_xcd = gridDim.x >> 3;              // one XCD's L2 sees all blocks of a segment (variant_kernels.hpp)
    const unsigned v = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (v >= p.n_work) return;
    const int seg = (int)(v / (unsigned)p.nquad);
    const int ind0A = ((int)(v % (unsigned)p.nquad) * WLOD2_BLOCKS + 2 * wave) * WAVE, ind0B = ind0A + WAVE;
    const bool activeA = ind0A < p.ind_count, activeB = ind0B < p.ind_count;
    const int2 td = p.tiles[seg];
    const ChrDev c = p.chrs[td.x];
    const int s0 = td.y;
    const int nwin = min(WSM_T, c.nloci - s0);            // windows of the segment that exist (scored or not)
    const int64_t G0 = c.loc_base + GOFF + s0;
    const int64_t lb = c.loc_base + s0;
    {   // score rows of the SNPs s0 .. and the windows' weights, coalesced, once
        if (!GL) {
            const int nrows = min(WSM_ROWS, c.nloci - s0);
            const double2 *src = reinterpret_cast<const double2 *>(wtab + G0 * 4);
            double2 *dst = reinterpret_cast<double2 *>(rows);
            for (int k = threadIdx.x; k < nrows * 2; k += WLOD_WAVES * WAVE) dst[k] = src[k];
        }
        const double *wsrc = rld + lb * WC;
        for (int k = threadIdx.x; k < nwin * WC; k += WLOD_WAVES * WAVE) wts[k] = wsrc[k];
    }
    // which windows hold a score: four ballots over the segment
    static_assert(WSM_T == 4 * WAVE, "four ballots");
    const uint64_t vm0 = __ballot(lane < nwin && p.valid[lb + lane] != 0);
    const uint64_t vm1 = __ballot(WAVE + lane < nwin && p.valid[lb + WAVE + lane] != 0);
    const uint64_t vm2 = __ballot(2 * WAVE + lane < nwin && p.valid[lb + 2 * WAVE + lane] != 0);
    const uint64_t vm3 = __ballot(3 * WAVE + lane < nwin && p.valid[lb + 3 * WAVE + lane] != 0);
    __syncthreads();
    if (!activeA) return;
    const int64_t colA = (int64_t)p.ind_begin + ind0A + lane;
    const int64_t colB = activeB ? colA + WAVE : colA;    // no second block: the first one again, results dropped
    const uint32_t *gA = packed + packed_index(G0 >> 4, colA, p.nwordrows);
    const uint32_t *gB = packed + packed_index(G0 >> 4, colB, p.nwordrows);
    const uint32_t shift = 2 * (uint32_t)(G0 & 15);
    const bool wholeA = ind0A + WAVE <= p.ind_count, wholeB = ind0B + WAVE <= p.ind_count;
    // group g reads the words g, g+1, g+2 of the lane's stream: two carried over, one requested a group ahead
    uint32_t a0 = 0, a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0;
    if (!GL) {
        a0 = gA[0]; a1 = gA[WAVE]; a2 = gA[2 * WAVE];
        b0 = gB[0]; b1 = gB[WAVE]; b2 = gB[2 * WAVE];
    }
    // GL: the lanes' own rows of the term matrix, row i of the segment at tA[i * 64] (rows past a chromosome's end exist:
    // GPAD_BACK; what they hold only reaches windows without a score)
    const double *tA = wtab + ((colA >> 6) * p.score_rows + G0) * WAVE + (colA & 63);
    const double *tB = wtab + ((colB >> 6) * p.score_rows + G0) * WAVE + (colB & 63);
    double sa[WC + 15], sb[WC + 15];    // the lanes' scores of a group's 16 + WC - 1 SNPs
    bool carried = false;               // GL: sa / sb [16 ..] are the next group's first WC - 1
    const int ngroups = (nwin + 15) >> 4;
#pragma unroll 1
    for (int g = 0; g < ngroups; g++) {
        const int sg = s0 + 16 * g;
        const uint64_t vmq = g < 4 ? vm0 : g < 8 ? vm1 : g < 12 ? vm2 : vm3;
        const uint32_t gm = (uint32_t)((vmq >> (16 * (g & 3))) & 0xFFFFu);
        uint32_t a3 = 0, b3 = 0;
        if (!GL) { a3 = gA[(g + 3) * WAVE]; b3 = gB[(g + 3) * WAVE]; }      // (the genotype array is padded far beyond)
        double acc[16], bcc[16];
        if (GL && gm == 0) carried = false;
        if (gm != 0) {
            const WsmGeno ga = wsm_align(a0, a1, a2, shift), gb = wsm_align(b0, b1, b2, shift);
            const double *rw = rows + 16 * g * 4;
            // the group's 16 * WC weights: lane k holds weight k (+ 64 m) -- one LDS read per 64 of them, then
            // v_readlane into scalar registers where a window uses one (a broadcast LDS read per weight kept the LDS
            // busier than anything else in the kernel: 160 reads per group at W = 10)
            constexpr int NWV = (16 * WC + WAVE - 1) / WAVE;
            double wv[NWV];
#pragma unroll
            for (int m = 0; m < NWV; m++) wv[m] = wts[16 * g * WC + min(lane + WAVE * m, 16 * WC - 1)];
            if (GL) {
                const double *ra = tA + (int64_t)(16 * g) * WAVE, *rb = tB + (int64_t)(16 * g) * WAVE;
                if (carried) {
#pragma unroll
                    for (int i = 0; i < WC - 1; i++) { sa[i] = sa[16 + i]; sb[i] = sb[16 + i]; }
#pragma unroll
                    for (int i = WC - 1; i < WC + 15; i++) {
                        sa[i] = __builtin_nontemporal_load(ra + i * WAVE);
                        sb[i] = __builtin_nontemporal_load(rb + i * WAVE);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < WC + 15; i++) {
                        sa[i] = __builtin_nontemporal_load(ra + i * WAVE);
                        sb[i] = __builtin_nontemporal_load(rb + i * WAVE);
                    }
                }
                carried = true;
            } else {
                // the lanes' scores of the group's 16 + WC - 1 SNPs, looked up as the windows need them
#pragma unroll
                for (int i = 0; i < WC + 15; i++) {
                    const uint32_t qa = ((i < 16 ? ga.lo : ga.hi) >> (2 * (i & 15))) & 3u;
                    const uint32_t qb = ((i < 16 ? gb.lo : gb.hi) >> (2 * (i & 15))) & 3u;
                    sa[i] = rw[i * 4 + qa];
                    sb[i] = rw[i * 4 + qb];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                double x = 0.0, y = 0.0;
#pragma unroll
                for (int j = 0; j < WC; j++) {
                    const int k = r * WC + j;
                    const double wk = wv[k / WAVE];
                    const double w = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(wk), k % WAVE),
                                                      __builtin_amdgcn_readlane(__double2loint(wk), k % WAVE));
                    const double pa = sa[r + j] * w, pb = sb[r + j] * w;
                    x = x + pa;
                    y = y + pb;
                }
                acc[r] = x;
                bcc[r] = y;
            }
        }
        if (sg + 16 <= c.nloci && wholeA) wlod_small_write<true>(acc, gm, c, p, out, mpatch, ind0A, sg, lane);
        else wlod_small_write<false>(acc, gm, c, p, out, mpatch, ind0A, sg, lane);
        if (activeB) {
            if (sg + 16 <= c.nloci && wholeB) wlod_small_write<true>(bcc, gm, c, p, out, mpatch, ind0B, sg, lane);
            else wlod_small_write<false>(bcc, gm, c, p, out, mpatch, ind0B, sg, lane);
        }
        a0 = a1; a1 = a2; a2 = a3;
        b0 = b1; b1 = b2; b2 = b3;
    }
}

// a kernel per window size: each with the registers its own unrolled group needs
template <int WC>
__global__ void __launch_bounds__(WLOD_WAVES * WAVE, WC <= 7 ? 4 : 3)
wlod_stream_small_kernel(const uint32_t *__restrict__ packed, const double *__restrict__ wtab,
                         const double *__restrict__ rld, double *__restrict__ out, WlodArgs p)
{
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    wlod_small_body<WC, false>(packed, wtab, rld, out, p, dyn);
}

// ... with per-genotype likelihoods: wtab is the scaled term matrix
template <int WC>
__global__ void __launch_bounds__(WLOD_WAVES * WAVE, WC <= 7 ? 4 : 3)
wlod_stream_small_gl_kernel(const uint32_t *__restrict__ packed, const double *__restrict__ terms,
                            const double *__restrict__ rld, double *__restrict__ out, WlodArgs p)
{
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    wlod_small_body<WC, true>(packed, terms, rld, out, p, dyn);
}

inline const void *wlod_stream_small_gl_fn(int W)
{
    switch (W) {
    case 2: return (const void *)wlod_stream_small_gl_kernel<2>;
    case 3: return (const void *)wlod_stream_small_gl_kernel<3>;
    case 4: return (const void *)wlod_stream_small_gl_kernel<4>;
    case 5: return (const void *)wlod_stream_small_gl_kernel<5>;
    case 6: return (const void *)wlod_stream_small_gl_kernel<6>;
    case 7: return (const void *)wlod_stream_small_gl_kernel<7>;
    case 8: return (const void *)wlod_stream_small_gl_kernel<8>;
    case 9: return (const void *)wlod_stream_small_gl_kernel<9>;
    case 10: return (const void *)wlod_stream_small_gl_kernel<10>;
    case 11: return (const void *)wlod_stream_small_gl_kernel<11>;
    case 12: return (const void *)wlod_stream_small_gl_kernel<12>;
    case 13: return (const void *)wlod_stream_small_gl_kernel<13>;
    case 14: return (const void *)wlod_stream_small_gl_kernel<14>;
    case 15: return (const void *)wlod_stream_small_gl_kernel<15>;
    default: return nullptr;
    }
}

inline const void *wlod_stream_small_fn(int W)
{
    switch (W) {
    case 2: return (const void *)wlod_stream_small_kernel<2>;
    case 3: return (const void *)wlod_stream_small_kernel<3>;
    case 4: return (const void *)wlod_stream_small_kernel<4>;
    case 5: return (const void *)wlod_stream_small_kernel<5>;
    case 6: return (const void *)wlod_stream_small_kernel<6>;
    case 7: return (const void *)wlod_stream_small_kernel<7>;
    case 8: return (const void *)wlod_stream_small_kernel<8>;
    case 9: return (const void *)wlod_stream_small_kernel<9>;
    case 10: return (const void *)wlod_stream_small_kernel<10>;
    case 11: return (const void *)wlod_stream_small_kernel<11>;
    case 12: return (const void *)wlod_stream_small_kernel<12>;
    case 13: return (const void *)wlod_stream_small_kernel<13>;
    case 14: return (const void *)wlod_stream_small_kernel<14>;
    case 15: return (const void *)wlod_stream_small_kernel<15>;
    default: return nullptr;
    }
}

} // namespace garlic
