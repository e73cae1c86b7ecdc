// ROH coverage counts straight from the genotypes: the window scores of calcLOD (src/garlic-roh.cpp:18-132) and the
// inWin[] loop of assembleROHWindows (src/garlic-roh.cpp:446-454) in one kernel.
//
//   inWin[l] = #{ windows w in (l - W, l] of this individual with score >= cutoff }
//
// (The shipped path is the pair of kernels at the END of this file -- lod_bits_kernel of feed_kernel.hpp + the counts
// from its bits --; the one-kernel forms below are what led there and stay selectable: GARLIC_COVERAGE_ONE_KERNEL.)
// GARLIC's final pass computes every window score of the chosen size (8 B per window), keeps them, and counts.  Here
// the scores never exist in memory: the chain of lod_feed_kernel (every wave a chain of its own, lane = individual,
// term rows through LDS rings, feed_kernel.hpp) produces a window's score in a register, one compare and a ballot
// turn it into a bit per individual, and the count is a sliding sum over the last W bits: 2 B per window leave the
// chip instead of 8 B out + 8 B in again + 2 B out (garlic_lod_windows + garlic_roh_coverage), and no score matrix
// (100 GB at 10M SNPs x 1250 individuals) has to be resident.
//
//   bits     32 <= W <= 224: the lane's own bits of the last 256 windows in eight registers (lod_coverage_kernel<true>,
//            below); other sizes: the wave's ballot masks (bit = individual) of the last W windows in an LDS ring of
//            the wave's own, the window leaving the count read back W steps later (one broadcast read);
//   counts   32 per tile and lane, packed two per dword, stored as four 16-byte pieces of the lane's row when the
//            layout allows (pitch_align a multiple of 8, interior tiles), one by one otherwise;
//   range    a run [a, b] of scored windows covers the SNPs a .. b + W - 1; runs are at least W - 1 windows apart
//            (whatever invalidates a window invalidates the W - 1 starts before it), so the regions of two runs never
//            overlap; SNPs no run covers are zeroed by fill_i16_ranges_kernel.
// Windows without a score never qualify here; the reference compares MISSING with the cutoff too, so the host only
// takes this path for cutoff > -9999 (and finite terms, and no -9999.0 sums: as for the thinned feed).
#pragma once
#include <type_traits>
#include "feed_kernel.hpp"
#include "cov_counts.hpp"

namespace garlic {

constexpr int COVF_MAX_W = 1024;

struct CovArgs {
    const uint32_t *packed;   // [nind_pad/64][nwordrows][64]
    const double *tab;        // [GOFF + nloci + pad][4]
    const FeedItem *items;    // (run, FEED_G blocks), longest first
    const ChrDev *chrs;       // out_base / out_pitch: the int16 count rows, per chromosome
    int16_t *out;
    int64_t nwordrows;
    int32_t ind_count, winsize, n_items, ring;   // ring: masks per wave (>= winsize + 32, a multiple of 32)
    int32_t vec_ok;                              // rows allow aligned 16-byte stores
    double cutoff;
    int32_t *next_item;       // [0] queue head, [1] workgroups that have left
};

struct CovRange { int32_t chr, lo, hi; };        // SNPs [lo, hi) of a chromosome no run covers

__global__ void __launch_bounds__(256)
fill_i16_ranges_kernel(const CovRange *__restrict__ ranges, const ChrDev *__restrict__ chrs, int nrows, int16_t *__restrict__ out)
{
    const CovRange r = ranges[blockIdx.x];
    const ChrDev c = chrs[r.chr];
    const int n = r.hi - r.lo;
    for (int row = blockIdx.y; row < nrows; row += gridDim.y) {
        int16_t *o = out + c.out_base + (int64_t)row * c.out_pitch + r.lo;
        for (int k = threadIdx.x; k < n; k += blockDim.x) o[k] = 0;
    }
}

// wave-uniform select / assign among the eight history dwords (registers cannot be indexed: a scalar branch chain)
__device__ __forceinline__ uint32_t covf_get(const uint32_t (&h)[8], int idx)
{
    switch (idx & 7) {
    case 0: return h[0];
    case 1: return h[1];
    case 2: return h[2];
    case 3: return h[3];
    case 4: return h[4];
    case 5: return h[5];
    case 6: return h[6];
    default: return h[7];
    }
}
__device__ __forceinline__ void covf_set(uint32_t (&h)[8], int idx, uint32_t v)
{
    switch (idx & 7) {
    case 0: h[0] = v; break;
    case 1: h[1] = v; break;
    case 2: h[2] = v; break;
    case 3: h[3] = v; break;
    case 4: h[4] = v; break;
    case 5: h[5] = v; break;
    case 6: h[6] = v; break;
    default: h[7] = v; break;
    }
}

// REGHIST (32 <= W <= COVF_REG_MAX_W): the lane's qualifying bits of the last 256 windows live in eight registers
// of its own (dword t & 7 = the 32 windows of tile t) instead of the wave's ballot masks in LDS: per tile ONE funnel
// shift lines up the 32 bits that leave the count (windows s0 - W ..), per window the work is a compare, a select,
// an OR into the tile's dword, a signed one-bit extract and a three-operand add -- no LDS traffic, no ballot, no
// 64-bit shift.  Tiles that lie inside (a, b] skip the in-run tests.  The next tile's genotype words are requested
// before this tile's arithmetic (loaded where they are used, every tile began with a trip to memory on the run's
// critical path: the compiler-scheduled tiles of the feed kernel run 3.8 x slower than its hand-scheduled loop for
// that reason).
constexpr int COVF_REG_MAX_W = 224;

template <bool REGHIST>
__global__ void __launch_bounds__(FEED_G * WAVE, 4)
lod_coverage_kernel(CovArgs p)
{
    // [0, GARLIC_FEED_LDS_TOTAL): the term-row rings of feed_kernel.hpp; behind them FEED_G rings of ballot masks
    extern __shared__ __attribute__((aligned(1024))) unsigned char cov_smem[];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t dma_off = (uint32_t)lane * 4u + (uint32_t)wave * 256u;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)cov_smem);
    uint64_t *ring = reinterpret_cast<uint64_t *>(cov_smem + GARLIC_FEED_LDS_TOTAL) + (size_t)wave * p.ring;
    const int W = p.winsize, R = p.ring;
    const double cutoff = p.cutoff;
    for (;;) {
        if (threadIdx.x == 0) *reinterpret_cast<int *>(cov_smem) = atomicAdd(p.next_item, 1);
        __syncthreads();
        const int item_idx = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int *>(cov_smem));
        __syncthreads();
        if (item_idx >= p.n_items) {
            if (threadIdx.x == 0) {
                __threadfence();
                if (atomicAdd(p.next_item + 1, 1) == (int)gridDim.x - 1) {
                    p.next_item[0] = 0;
                    p.next_item[1] = 0;
                }
            }
            return;
        }
        const FeedItem *it = p.items + item_idx;
        const ChrDev c = p.chrs[it->chr];
        const int a = it->a, b = it->b, prio = it->prio;
        if (prio >= 3) __builtin_amdgcn_s_setprio(3);
        else if (prio == 2) __builtin_amdgcn_s_setprio(2);
        else if (prio == 1) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
        const int ind0 = __builtin_amdgcn_readfirstlane(it->ind0[wave]);
        const bool active = ind0 >= 0;
        const int row = (active && ind0 + lane < p.ind_count) ? ind0 + lane : -1;
        const int64_t col0 = active ? ind0 : 0;
        const uint32_t *gcol = p.packed + packed_index(0, col0 + lane, p.nwordrows);
        const int64_t Gbase = c.loc_base + GOFF;
        const int first = a & ~(TILE - 1);
        const int last = min(b + W - 1, c.nloci - 1);           // last SNP this run covers
        const int ntiles = ((last - first) >> 5) + 1;
        const double *lead_chunks = p.tab + (Gbase + first + W - 1) * 4;
        const double *trail_chunks = p.tab + (Gbase + first - 1) * 4;
        for (int t = 0; t < FEED_AHEAD; t++) {
            const uint32_t slot = (uint32_t)((t + 3) & 3) * 1024u + (uint32_t)wave * 256u;
            feed_dma_quarter(lds0 + GARLIC_FEED_LDS_LEAD + slot, lead_chunks + (int64_t)t * 128, dma_off);
            feed_dma_quarter(lds0 + GARLIC_FEED_LDS_TRAIL + slot, trail_chunks + (int64_t)t * 128, dma_off);
        }
        // first window of the run: its first W - 1 terms left to right (garlic-roh.cpp:57-71), as in lod_feed_kernel
        double acc = 0.0;
        if (active) {
            int l = a;
            const int lend = a + W - 1;
            while (l < lend) {
                const int64_t G = Gbase + l;
                const int sh = 2 * (int)(G & 15);
                const uint32_t *wp = gcol + (G >> 4) * WAVE;
                const uint32_t w0 = wp[0], w1 = wp[WAVE], w2 = wp[2 * WAVE];
                const uint32_t al[2] = {__builtin_amdgcn_alignbit(w1, w0, sh), __builtin_amdgcn_alignbit(w2, w1, sh)};
                const int n = min(32, lend - l);
                double t[32];
#pragma unroll
                for (int q = 0; q < 32; q++) {
                    const uint32_t g = (al[q >> 4] >> (2 * (q & 15))) & 3u;
                    t[q] = p.tab[(G + min(q, n - 1)) * 4 + ((q < n) ? g : 3u)];
                }
#pragma unroll
                for (int q = 0; q < 32; q++) acc += (q < n) ? t[q] : 0.0;
                l += n;
            }
        }
        const int64_t Glead = Gbase + first + W - 1, Gtrail = Gbase + first - 1;
        const int sh_lead = 2 * (int)(Glead & 15), sh_trail = 2 * (int)(Gtrail & 15);
        int16_t *const orow = p.out + c.out_base + (int64_t)max(row, 0) * c.out_pitch;
        int cnt = 0;
        int wr = 0;                              // (LDS ring) slot of window `first + 32 k`
        int rd = (R - (W % R)) % R;              // slot of the window W steps back: (wr - W) mod R
        uint32_t hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        // the lane's genotype words: tile k reads word rows 2 k .. 2 k + 2 of each stream; those of tile k + 2 are
        // requested inside tile k (two tiles of arithmetic cover the trip to memory)
        const uint32_t *lw = gcol + (Glead >> 4) * WAVE, *tw = gcol + (Gtrail >> 4) * WAVE;
        uint32_t l0 = lw[0], l1 = lw[WAVE], l2 = lw[2 * WAVE], l3 = lw[3 * WAVE], l4 = lw[4 * WAVE];
        uint32_t t0 = tw[0], t1 = tw[WAVE], t2 = tw[2 * WAVE], t3 = tw[3 * WAVE], t4 = tw[4 * WAVE];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int k = 0; k < ntiles; k++) {
            feed_barrier();
            {
                const uint32_t slot = (uint32_t)((k + FEED_AHEAD + 3) & 3) * 1024u + (uint32_t)wave * 256u;
                feed_dma_quarter(lds0 + GARLIC_FEED_LDS_LEAD + slot, lead_chunks + (int64_t)(k + FEED_AHEAD) * 128, dma_off);
                feed_dma_quarter(lds0 + GARLIC_FEED_LDS_TRAIL + slot, trail_chunks + (int64_t)(k + FEED_AHEAD) * 128, dma_off);
            }
            if (active) {
                const int s0 = first + k * TILE;
                const uint32_t lead_w[2] = {__builtin_amdgcn_alignbit(l1, l0, sh_lead), __builtin_amdgcn_alignbit(l2, l1, sh_lead)};
                const uint32_t trail_w[2] = {__builtin_amdgcn_alignbit(t1, t0, sh_trail), __builtin_amdgcn_alignbit(t2, t1, sh_trail)};
                // the words of tile k + 2 (word rows 2 k + 5, 2 k + 6): requested now, looked at two tiles later
                lw += 2 * WAVE; tw += 2 * WAVE;
                l0 = l2; l1 = l3; l2 = l4; t0 = t2; t1 = t3; t2 = t4;
                l3 = lw[3 * WAVE]; l4 = lw[4 * WAVE];
                t3 = tw[3 * WAVE]; t4 = tw[4 * WAVE];
                const uint32_t lead_rows = lds0 + GARLIC_FEED_LDS_LEAD + (uint32_t)((k + 3) & 3) * 1024u;
                const uint32_t trail_rows = lds0 + GARLIC_FEED_LDS_TRAIL + (uint32_t)((k + 3) & 3) * 1024u;
                uint32_t pk[16];
                const bool interior = s0 > a && s0 + TILE - 1 <= b;       // wave-uniform
                if (REGHIST) {
                    // bits of the windows s0 - W + j, j = 0 .. 31: position 32 k - W + j in the run's bit string
                    const int rel = 32 * k - W;                          // may be negative: those dwords are still zero
                    const int dA = rel >> 5, r = rel & 31;               // (arithmetic shift: floor)
                    const uint32_t F = __builtin_amdgcn_alignbit(covf_get(hist, dA + 1), covf_get(hist, dA), (uint32_t)r);
                    uint32_t hnew = 0;
                    auto tile = [&](auto edge_tag) {
                        constexpr bool EDGE = decltype(edge_tag)::value;
#pragma unroll
                        for (int bq = 0; bq < 4; bq++) {
                            double tin[8], tout[8];
#pragma unroll
                            for (int i = 0; i < 8; i++) {
                                const int j = 8 * bq + i;
                                tin[i] = feed_lds_double(feed_addr(lead_w[j >> 4], j & 15, lead_rows) + (uint32_t)j * 32u);
                                tout[i] = feed_lds_double(feed_addr(trail_w[j >> 4], j & 15, trail_rows) + (uint32_t)j * 32u);
                            }
#pragma unroll
                            for (int i = 0; i < 8; i++) {
                                const int j = 8 * bq + i, s = s0 + j;
                                const bool in_run = !EDGE || (s >= a && s <= b);
                                const double to = (!EDGE || (s > a && s <= b)) ? tout[i] : 0.0;
                                const double ti = in_run ? tin[i] : 0.0;
                                acc = (acc - to) + ti;   // two roundings, as garlic-roh.cpp:98-100
                                const uint32_t qbit = (in_run && acc >= cutoff) ? 1u : 0u;      // NaN >= x is false
                                hnew |= qbit << j;
                                const int outneg = __builtin_amdgcn_sbfe(F, (uint32_t)j, 1u);   // 0 or -1
                                cnt = cnt + (int)qbit + outneg;
                                const uint32_t v = (uint32_t)cnt & 0xFFFFu;
                                if (j & 1) pk[j >> 1] |= v << 16;
                                else pk[j >> 1] = v;
                            }
                        }
                    };
                    if (interior) tile(std::false_type{});
                    else tile(std::true_type{});
                    covf_set(hist, k, hnew);
                } else {
#pragma unroll
                    for (int bq = 0; bq < 4; bq++) {
                        double tin[8], tout[8];
#pragma unroll
                        for (int i = 0; i < 8; i++) {
                            const int j = 8 * bq + i;
                            tin[i] = feed_lds_double(feed_addr(lead_w[j >> 4], j & 15, lead_rows) + (uint32_t)j * 32u);
                            tout[i] = feed_lds_double(feed_addr(trail_w[j >> 4], j & 15, trail_rows) + (uint32_t)j * 32u);
                        }
                        // the masks of the eight windows leaving the count, read together; W < 8: they include masks
                        // this batch writes, read one by one below
                        uint64_t oldm[8];
                        if (W >= 8) {
#pragma unroll
                            for (int i = 0; i < 8; i++) {
                                int r2 = rd + i;
                                r2 = r2 >= R ? r2 - R : r2;
                                oldm[i] = ring[r2];
                            }
                        }
#pragma unroll
                        for (int i = 0; i < 8; i++) {
                            const int j = 8 * bq + i, s = s0 + j;
                            const bool in_run = s >= a && s <= b;
                            const double to = (s > a && s <= b) ? tout[i] : 0.0;
                            const double ti = in_run ? tin[i] : 0.0;
                            acc = (acc - to) + ti;   // two roundings, as garlic-roh.cpp:98-100
                            const bool q = in_run && acc >= cutoff;      // NaN >= x is false
                            const uint64_t mask = __ballot(q);
                            // the window leaving (s - W, s]: window s - W, if this run holds it
                            uint64_t old = W >= 8 ? oldm[i] : ring[rd];
                            if (s - W < a) old = 0;
                            if (lane == 0) ring[wr] = mask;
                            wr = (wr + 1 == R) ? 0 : wr + 1;
                            rd = (rd + 1 == R) ? 0 : rd + 1;
                            cnt += (int)q - (int)((old >> lane) & 1);
                            const uint32_t v = (uint32_t)cnt & 0xFFFFu;
                            if (j & 1) pk[j >> 1] |= v << 16;
                            else pk[j >> 1] = v;
                        }
                    }
                }
                const bool vec = p.vec_ok && s0 >= a && s0 + TILE - 1 <= last;       // wave-uniform
                if (vec) {
                    if (row >= 0) {
                        uint4 *o = reinterpret_cast<uint4 *>(orow + s0);
#pragma unroll
                        for (int u = 0; u < 4; u++) o[u] = make_uint4(pk[4 * u], pk[4 * u + 1], pk[4 * u + 2], pk[4 * u + 3]);
                    }
                    // Everything up to the previous tile's requests has to have landed (its quarters of the chunks of
                    // tile k + 2, the words of tile k + 1); this tile's own -- two chunk quarters, four words, four
                    // stores -- stay in flight: a run's tiles are its wave's critical path.  (The chunks of tile k + 3
                    // are first read three tiles on; the barrier of tile k + 2 follows this wave's wait for them.)
                    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                    continue;
                }
                if (row >= 0) {
#pragma unroll
                    for (int j = 0; j < TILE; j++) {
                        const int s = s0 + j;
                        if (s >= a && s <= last) orow[s] = (int16_t)((pk[j >> 1] >> (16 * (j & 1))) & 0xFFFFu);
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_setprio(0);
    }
}

// ---- two kernels: bits, then counts.  lod_bits_kernel (feed_kernel.hpp) is the thinned feed's hand-scheduled chain
// with a compare and an add-with-carry per window instead of the sampled stores: it leaves ONE BIT per window and
// individual (10.1 instructions per window on the run's critical path against 18 for the compiler-scheduled
// one-kernel form above); the counts are then a pass with no chain in it at all:
//   inWin[l] = #bits in (l - W, l]
// a launch of its own over every (individual, 32-SNP word): cov_counts_word of cov_counts.hpp per thread
__global__ void __launch_bounds__(256)
cov_counts_from_bits_kernel(const uint32_t *__restrict__ bits, const ChrDev *__restrict__ bchrs,
                            const ChrDev *__restrict__ ochrs, const int32_t *__restrict__ word_base, int nchr, int W,
                            int nind, int vec_ok, int16_t *__restrict__ out)
{
    __shared__ uint4 xpose[4][COV_XPOSE_SLOTS];
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x >> 6;
    const int g = (int)(blockIdx.x * blockDim.x + threadIdx.x);      // word column over all chromosomes
    const bool live = g < word_base[nchr];
    const int chr = live ? cov_word_chr(word_base, nchr, g) : 0;
    const ChrDev bc = bchrs[chr], oc = ochrs[chr];
    const int t = live ? g - word_base[chr] : 0, nwords = (bc.nloci + 31) >> 5;
    const bool whole = live && vec_ok && 32 * t + 32 <= oc.nloci;
    // all 64 threads of the wave on whole words of one chromosome: their 4 KB are contiguous in the row
    const int chr0 = __builtin_amdgcn_readfirstlane(chr);
    const bool wave_whole = __ballot(whole && chr == chr0) == ~(uint64_t)0;
    // COV_ITEM_ROWS individuals per workgroup: one workgroup per 256 words and individual is 1.5 M workgroups at
    // 10M SNPs x 1250 and ran at the dispatcher's pace (3.8 TB/s)
    const int row_end = min(nind, ((int)blockIdx.y + 1) * COV_ITEM_ROWS);
    for (int row = (int)blockIdx.y * COV_ITEM_ROWS; row < row_end; row++)
        cov_counts_word(bits + bc.out_base + (int64_t)row * bc.out_pitch, nwords, t, live, W,
                        out + oc.out_base + (int64_t)row * oc.out_pitch, oc.nloci, whole, wave_whole, xpose[wave], lane);
}

} // namespace garlic
