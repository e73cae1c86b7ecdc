// Thinned LOD scores for the KDE feed (the explore / auto-winsize flows, src/garlic-roh.cpp:726-751, 798-837,
// 881-920 -> convertWinData2DoubleData, src/garlic-data.cpp:2026-2069: only the windows at chromosome-local
// loci 0, step, 2*step, .. are ever looked at), and the same chains leaving one bit per window (score >= cutoff)
// for the final pass (coverage_kernel.hpp, roh_segments_kernel.hpp).
//
// With no score stream to write, the window recurrence (src/garlic-roh.cpp:92-100) is all there is, and the
// four-role kernel of lod_kernels.hpp -- three helper waves feeding ONE chain wave per CU, built for the
// store-bound full output -- leaves the machine idle.  Here every wave is a chain of its own and a CU runs
// sixteen of them:
//
//   work item   one run of valid windows x FEED_G 64-individual blocks = one workgroup of FEED_G waves (lane =
//               individual); persistent workgroups (three per CU: 42 KB of LDS each) pull items longest run first; the waves of the
//               longest runs raise their issue priority (s_setprio): the run length x the pace of one wave is
//               the kernel's critical path, everything shorter fills the issue slots they leave;
//   interior    tiles of 32 windows, GARLIC_FEED_UNROLL of them per iteration of the hand-scheduled loop of
//               tools/gen_feed_asm.py (feed_loop_gfx950.inc; round 4: 5.8 instructions per window and lane, 7.6 with
//               the coverage bit, 1.5 vector-memory instructions per tile and wave): per window ONE look-up -- {t_out,
//               t_in} of the lane's genotype pair, a ds_read_b128 from the window's 16-entry pair table -- whose offset
//               comes out of a word of 4-bit genotype-pair codes by one SDWA instruction, and the two dependent adds
//               acc = (acc - t_out) + t_in.  The pair tables (8 KB per tile, a ring of four in LDS) are built by the
//               workgroup's waves themselves, a quarter each, from the tile's raw term rows, which the waves take turns
//               to bring in as two 1-KB loads; one s_barrier per tile is the rings' protocol;
//   genotypes   the lanes' packed words, 4 word rows (1 KB) per load and stream, turned lane-wise through 1 KB of LDS
//               into register rings; the leaving stream is a second, cache-served read of the same words;
//   edges       a run's first tile, its last ones, the tiles in front of the loop's first aligned one and everything
//               of a shard that does not start on a block boundary go through the compiler-generated tile below,
//               which takes its terms from the term table in memory and needs no ring;
//   samples     a sampled locus goes straight from the lanes into a [row][column] matrix per chromosome, eight
//               consecutive samples of a lane as one 64-byte piece (edges: 8 B each).
//               The host chooses rows and columns: row = individual, column = locus / step gives the thinned score
//               matrix; row = position in the caller's individual list, column = rank of the sample among the
//               chromosome's scored samples gives the KDE feed itself (convertWinData2DoubleData's order: the mask
//               is the same for every individual, so the rank of a sample is known before anything is computed);
//   bits        (lod_bits_kernel) a tile's 32 bits are one dword per lane; the loop starts at a tile index that is a
//               multiple of eight and stores eight tiles' dwords as one aligned 32-byte piece per lane.
#pragma once
#include "lod_kernels.hpp"
#include "cov_counts.hpp"
#include "feed_loop_gfx950.inc"

namespace garlic {

constexpr int FEED_G = 4;                 // waves (64-individual blocks) per workgroup: each builds a quarter of a tile's pair table

struct FeedItem {
    int32_t chr, a, b;        // run of valid windows [a, b] (chromosome-local)
    int32_t prio;             // 0..3: issue priority of the item's waves (long runs first)
    int32_t ind0[FEED_G];     // first individual of each wave's block (relative to ind_begin); -1: none
    int32_t col0;             // column (in the chromosome's rows of the sample matrix) of the run's first sampled locus
    int32_t pad[3];
};

struct FeedArgs {
    const uint32_t *packed;   // [nind_pad/64][nwordrows][64]
    const double *tab;        // [GOFF + nloci + pad][4]
    const FeedItem *items;
    const ChrDev *chrs;       // out_base / out_pitch: the sample matrix, [row][column] per chromosome
    double *out;
    const int32_t *row_map;   // row of each individual of the call in the sample matrix, -1: none; NULL: row = individual
    int64_t nwordrows;
    int32_t ind_begin, ind_count, winsize, n_items, thin_step;
    int32_t use_asm;          // 0: every tile through the compiler-generated path (GARLIC_FEED_NO_ASM)
    int32_t *next_item;       // [0] queue head, [1] workgroups that have left (zero at launch, reset by the last)
    int64_t *trace;           // optional (GARLIC_TRACE): per item {workgroup, begin, tiles begin, end} in 100 MHz ticks + shader clocks
    // lod_bits_kernel only: instead of sampled scores ONE BIT per window and individual, score >= cutoff, 32 windows
    // (a tile) per dword: `out` is then a uint32 matrix, chrs[].out_base / out_pitch in dwords, row = individual,
    // column = chromosome-local tile index (locus / 32); zeroed by the caller (edge tiles OR their bits in)
    double cutoff;
    // lod_bits_kernel, optional (cnt_order != NULL): the sliding counts from the bits in the same launch.  The queue
    // continues behind the chain items with n_cnt_items count items (cov_counts.hpp), chromosomes in the order their
    // chains finish; every chain item adds one to chr_done[its chromosome] when its bits are in memory, a count item
    // waits for chr_done[c] == chr_need[c].  Queue order makes that safe: whoever holds a count item knows every chain
    // item has been taken by a workgroup that is running and waits for nobody.  The longest runs' chains are the kernel's
    // critical path (one wave's pace x the run length) and leave most of the chip idle: the counts of every chromosome
    // that is complete fill it.
    const int32_t *cnt_order; // [nchr] chromosomes in queue order (NULL: no count items); chromosome cnt_order[k] owns the
    const int32_t *cnt_base;  // [nchr + 1] count items cnt_base[k] .. cnt_base[k + 1] - 1: word chunk major, row group minor
    const ChrDev *cnt_chrs;   // out_base / out_pitch / nloci of the int16 count rows
    int16_t *cnt_out;
    int32_t *chr_done;        // [nchr], zero at launch
    const int32_t *chr_need;  // [nchr]: chain items per chromosome
    int32_t n_cnt_items, n_cnt_chr, cnt_vec_ok;
    int32_t *cnt_timeout;     // set to 1 if a count item's wait ran out of its poll budget (the counts are then not to be used)
};

constexpr int FEED_BITS_LDS = 1024 + 4 * COV_XPOSE_SLOTS * 16;   // lod_bits_kernel: the count items' transpose buffers (one per wave) behind the misc words

__device__ __forceinline__ void feed_barrier()
{   // s_barrier alone: __syncthreads() would also wait for every outstanding memory request
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// wave-uniform values for the "s" operands of the inline assembly (hipcc passes a VGPR where it has not proven uniformity)
__device__ __forceinline__ uint32_t feed_uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T> __device__ __forceinline__ T *feed_uni(T *ptr)
{
    const uint64_t u = reinterpret_cast<uint64_t>(ptr);
    return reinterpret_cast<T *>(((uint64_t)feed_uni((uint32_t)(u >> 32)) << 32) | feed_uni((uint32_t)u));
}

template <bool BITS>
__device__ __forceinline__ void lod_feed_body(const FeedArgs &p)
{
    // one LDS object at offset 0: the hand-scheduled loop addresses the rings with absolute offsets
    __shared__ __attribute__((aligned(1024))) unsigned char smem[BITS && FEED_BITS_LDS > GARLIC_FEED_LDS_TOTAL ? FEED_BITS_LDS : GARLIC_FEED_LDS_TOTAL];
    static_assert((GARLIC_FEED_UNROLL & (GARLIC_FEED_UNROLL - 1)) == 0, "tiles per iteration: a power of two");
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // LDS address of smem: 0, the kernel's only LDS object (taking it here also keeps the whole array allocated: most of
    // it is only ever touched through the integer addresses below and the loop's immediates)
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem);
    const int W = p.winsize, step = p.thin_step;
    int fenced_chr = -1;      // count items: the chromosome this workgroup last waited for (its bits are visible here)
    for (;;) {
        if (threadIdx.x == 0) *reinterpret_cast<int *>(smem) = atomicAdd(p.next_item, 1);
        __syncthreads();
        const int item_idx = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int *>(smem));
        __syncthreads();   // (also: every wave is done with the previous item's rings)
        if (BITS && item_idx >= p.n_items && p.cnt_order && item_idx < p.n_items + p.n_cnt_items) {
            // ---- a count item: COV_ITEM_WORDS words x COV_ITEM_ROWS individuals of a chromosome whose chains are done
            const int q = item_idx - p.n_items;
            int lo = 0, hi = p.n_cnt_chr - 1;          // the last k with cnt_base[k] <= q
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (p.cnt_base[mid] <= q) lo = mid;
                else hi = mid - 1;
            }
            const int chr = __builtin_amdgcn_readfirstlane(p.cnt_order[lo]);
            const int ql = q - p.cnt_base[lo], nrg = (p.ind_count + COV_ITEM_ROWS - 1) / COV_ITEM_ROWS;
            const int word0 = ql / nrg * COV_ITEM_WORDS, row0 = ql % nrg * COV_ITEM_ROWS;
            const int nrows = min(COV_ITEM_ROWS, p.ind_count - row0);
            if (chr != fenced_chr) {
                if (threadIdx.x == 0) {
                    const int need = p.chr_need[chr];
                    int polls = 0;
                    while (__hip_atomic_load(p.chr_done + chr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                        __builtin_amdgcn_s_sleep(127);
                        // (seconds: a chain that never finishes must not hang the device; once one wait has given up
                        // nobody waits any more and the host reports the call as failed)
                        if (++polls > (1 << 21) || __hip_atomic_load(p.cnt_timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                            __hip_atomic_store(p.cnt_timeout, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
#ifndef GARLIC_COVOV_ABL_NOACQ
                    __threadfence();     // acquire: the other XCDs' bits (their L2s were written back before chr_done moved)
#endif
                }
                __syncthreads();
                fenced_chr = chr;
            }
            const ChrDev bc = p.chrs[chr], oc = p.cnt_chrs[chr];
            const int nwords = (bc.nloci + 31) >> 5;
            const int t = word0 + (int)threadIdx.x;
            const bool live = t < nwords;
            const bool whole = live && p.cnt_vec_ok && 32 * t + 32 <= oc.nloci;
            const bool wave_whole = __ballot(whole) == ~(uint64_t)0;
            uint4 *xw = reinterpret_cast<uint4 *>(smem + 1024) + wave * COV_XPOSE_SLOTS;
            const uint32_t *bits = reinterpret_cast<const uint32_t *>(p.out);
#ifdef GARLIC_COVOV_ABL_NOWORK      // (timing experiment: the count items wait and do nothing)
            if (p.cnt_out) continue;
#endif
            for (int r = 0; r < nrows; r++) {
                const int row = row0 + r;
                cov_counts_word(bits + bc.out_base + (int64_t)row * bc.out_pitch, nwords, live ? t : 0, live, W,
                                p.cnt_out + oc.out_base + (int64_t)row * oc.out_pitch, oc.nloci, whole, wave_whole, xw, lane);
            }
            continue;
        }
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
        if (p.trace && threadIdx.x == 0) {
            p.trace[8 * item_idx + 0] = blockIdx.x;
            p.trace[8 * item_idx + 1] = wall_clock64();
            p.trace[8 * item_idx + 4] = clock64();
        }
        const FeedItem *it = p.items + item_idx;
        const ChrDev c = p.chrs[it->chr];
        const int a = it->a, b = it->b, prio = it->prio;
        if (prio >= 3) __builtin_amdgcn_s_setprio(3);
        else if (prio == 2) __builtin_amdgcn_s_setprio(2);
        else if (prio == 1) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
        const int ind0 = __builtin_amdgcn_readfirstlane(it->ind0[wave]);   // (indexed in memory: a local copy of the item would live in scratch)
        const bool active = ind0 >= 0;
        // the lane's row in the sample matrix
        int row = -1;
        if (active && ind0 + lane < p.ind_count) row = p.row_map ? p.row_map[ind0 + lane] : ind0 + lane;
        const int64_t col0 = (int64_t)p.ind_begin + (active ? ind0 : 0);
        const uint32_t *gcol = p.packed + packed_index(0, col0 + lane, p.nwordrows);
        const int64_t Gbase = c.loc_base + GOFF;

        const int first = a & ~(TILE - 1);
        const int ntiles = ((b - first) >> 5) + 1;
        // term rows {lod(0), lod(1), lod(2), +0.0}: window s takes SNP s + W - 1 in and SNP s - 1 out
        const double *lead_rows = p.tab + (Gbase + first + W - 1) * 4;
        const double *trail_rows = p.tab + (Gbase + first - 1) * 4;
        // (pad rows behind the table and the packed panel make the loop's requests past the run's last tile harmless)

        // ---- first window of the run: its first W-1 terms left to right (garlic-roh.cpp:57-71); the W-th
        //      enters in the first tile.  32 SNPs per round, every load of a round before its first add.
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

        // ---- streams: entering SNP of window s is s + W - 1, leaving SNP s - 1
        const int64_t Glead = Gbase + first + W - 1, Gtrail = Gbase + first - 1;
        const int sh_lead = 2 * (int)(Glead & 15), sh_trail = 2 * (int)(Gtrail & 15);
        // thinned output: next sampled locus at or after a; its column in the block's rows
        int next = (a + step - 1) / step * step;
        double *const out_chr = p.out + c.out_base;
        int col = it->col0;
        if (p.trace && threadIdx.x == 0) {
            p.trace[8 * item_idx + 2] = wall_clock64();
            p.trace[8 * item_idx + 5] = clock64();
        }

        // interior tiles 1 .. k_int (all 32 windows inside (a, b]); the hand-scheduled loop takes GARLIC_FEED_UNROLL n of
        // them from tile k0 on -- bits: the first tile whose dword sits at a multiple of eight in the lanes' rows, so that
        // an iteration's eight dwords are one aligned 32-byte piece
        const int k_int = (b - first - (TILE - 1)) >> 5;
        const int k0 = BITS ? GARLIC_FEED_UNROLL - ((first >> 5) & (GARLIC_FEED_UNROLL - 1)) : 1;
        const bool asm_ok = p.use_asm && lds0 == 0 && (col0 & 63) == 0 && W + GARLIC_FEED_REACH <= GPAD_BACK &&
                            (!BITS || (((c.out_pitch | c.out_base) & 7) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 31) == 0));
        const int niter = (asm_ok && k_int >= k0) ? (k_int - k0 + 1) / GARLIC_FEED_UNROLL : 0;
        for (int k = 0; k < ntiles; k++) {
            if (k == k0 && niter > 0) {
                const int s0 = first + k0 * TILE;
                const uint32_t *blk = p.packed + packed_index(0, col0, p.nwordrows);     // the block's word rows (lane 0)
                const uint32_t *plw = blk + ((Glead + (int64_t)k0 * TILE) >> 4) * WAVE;    // word row 0 of the loop's first tile
                const uint32_t *ptw = blk + ((Gtrail + (int64_t)k0 * TILE) >> 4) * WAVE;
                const double *ptl = lead_rows + (int64_t)k0 * TILE * 4, *ptt = trail_rows + (int64_t)k0 * TILE * 4;
                const double *out_next = out_chr + col;
                uint32_t next_rel = (uint32_t)(next - s0);
                if (BITS) {
                    // the dword of the loop's first tile in row 0 of the bit matrix; rows are out_pitch dwords apart
                    const uint32_t *bits_next = reinterpret_cast<const uint32_t *>(p.out) + c.out_base + (s0 >> 5);
                    const uint32_t *bits_out;
                    const uint64_t cutbits = __builtin_bit_cast(uint64_t, p.cutoff);
                    const uint64_t cut = ((uint64_t)feed_uni((uint32_t)(cutbits >> 32)) << 32) | feed_uni((uint32_t)cutbits);
                    asm volatile(GARLIC_FEED_BITS_LOOP_ASM
                                 : [acc] "+v"(acc), [next_out] "=s"(next_rel), [out_out] "=s"(bits_out)
                                 : [wave] "s"(wave), [lane] "v"(lane), [active] "s"(feed_uni(active ? 1u : 0u)), [plw] "s"(feed_uni(plw)),
                                   [ptw] "s"(feed_uni(ptw)), [ptl] "s"(feed_uni(ptl)), [ptt] "s"(feed_uni(ptt)),
                                   [out] "s"(feed_uni(bits_next)), [next] "s"(feed_uni(next_rel)), [step] "s"(feed_uni((uint32_t)step)),
                                   [shl] "s"(feed_uni((uint32_t)sh_lead)), [sht] "s"(feed_uni((uint32_t)sh_trail)),
                                   [row] "v"(row), [pitch8] "s"(feed_uni((uint32_t)(c.out_pitch * 4))),
                                   [niter] "s"(feed_uni((uint32_t)niter)), [cut] "s"(cut)
                                 : GARLIC_FEED_LOOP_CLOBBERS);
                } else {
                    asm volatile(GARLIC_FEED_LOOP_ASM
                                 : [acc] "+v"(acc), [next_out] "=s"(next_rel), [out_out] "=s"(out_next)
                                 : [wave] "s"(wave), [lane] "v"(lane), [active] "s"(feed_uni(active ? 1u : 0u)), [plw] "s"(feed_uni(plw)),
                                   [ptw] "s"(feed_uni(ptw)), [ptl] "s"(feed_uni(ptl)), [ptt] "s"(feed_uni(ptt)),
                                   [out] "s"(feed_uni(out_next)), [next] "s"(feed_uni(next_rel)), [step] "s"(feed_uni((uint32_t)step)),
                                   [shl] "s"(feed_uni((uint32_t)sh_lead)), [sht] "s"(feed_uni((uint32_t)sh_trail)),
                                   [row] "v"(row), [pitch8] "s"(feed_uni((uint32_t)(c.out_pitch * 8))),
                                   [niter] "s"(feed_uni((uint32_t)niter))
                                 : GARLIC_FEED_LOOP_CLOBBERS);
                    next = first + (k + GARLIC_FEED_UNROLL * niter) * TILE + (int)next_rel;
                    col = (int)(out_next - out_chr);
                }
                k += GARLIC_FEED_UNROLL * niter;
                if (k >= ntiles) break;
            }
            if (active) {
                const int s0 = first + k * TILE;
                const uint32_t *lw = gcol + ((Glead + (int64_t)k * TILE) >> 4) * WAVE;
                const uint32_t *tw = gcol + ((Gtrail + (int64_t)k * TILE) >> 4) * WAVE;
                const uint32_t l0 = lw[0], l1 = lw[WAVE], l2 = lw[2 * WAVE];
                const uint32_t t0 = tw[0], t1 = tw[WAVE], t2 = tw[2 * WAVE];
                const uint32_t lead_w[2] = {__builtin_amdgcn_alignbit(l1, l0, sh_lead), __builtin_amdgcn_alignbit(l2, l1, sh_lead)};
                const uint32_t trail_w[2] = {__builtin_amdgcn_alignbit(t1, t0, sh_trail), __builtin_amdgcn_alignbit(t2, t1, sh_trail)};
                const double *tl = lead_rows + (int64_t)k * TILE * 4, *tt = trail_rows + (int64_t)k * TILE * 4;
                uint32_t tile_bits = 0;
#pragma unroll
                for (int bq = 0; bq < 4; bq++) {
                    double tin[8], tout[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const int j = 8 * bq + i;
                        tin[i] = tl[4 * j + ((lead_w[j >> 4] >> (2 * (j & 15))) & 3u)];
                        tout[i] = tt[4 * j + ((trail_w[j >> 4] >> (2 * (j & 15))) & 3u)];
                    }
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const int s = s0 + 8 * bq + i;
                        // the first window of a run is a plain sum (no leaving term); steps outside [a, b] leave the
                        // accumulator as it is (x - 0.0 + 0.0 == x: it starts at +0.0 and never becomes -0.0)
                        const double to = (s > a && s <= b) ? tout[i] : 0.0;
                        const double ti = (s >= a && s <= b) ? tin[i] : 0.0;
                        acc = (acc - to) + ti;   // two roundings, as garlic-roh.cpp:98-100
                        if (BITS) {
                            if (s >= a && s <= b && acc >= p.cutoff) tile_bits |= 1u << (8 * bq + i);      // NaN >= x is false
                        } else if (s == next && s <= b) {
                            if (row >= 0) out_chr[(int64_t)row * c.out_pitch + col] = acc;
                            col++;
                            next += step;
                        }
                    }
                }
                // (a tile at a run's edge may hold another run's windows too when W < 32: OR, into the zeroed matrix)
                if (BITS && row >= 0 && tile_bits)
                    atomicOr(reinterpret_cast<uint32_t *>(p.out) + c.out_base + (int64_t)row * c.out_pitch + (s0 >> 5), tile_bits);
            }
        }
        if (p.trace && threadIdx.x == 0) {
            p.trace[8 * item_idx + 3] = wall_clock64();
            p.trace[8 * item_idx + 6] = clock64();
            p.trace[8 * item_idx + 7] = ntiles;
        }
        __builtin_amdgcn_s_setprio(0);
        if (BITS && p.cnt_order) {
            // the item's bits are in memory (every wave has waited for its stores) before its chromosome's count moves:
            // release at device scope, the count items may run on another XCD
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                atomicAdd(p.chr_done + p.items[item_idx].chr, 1);
            }
        }
    }
}

__global__ void __launch_bounds__(FEED_G * WAVE, GARLIC_FEED_WG_PER_CU)
lod_feed_kernel(FeedArgs p)
{
    lod_feed_body<false>(p);
}

// the same chains leaving one bit per window and individual (score >= cutoff) instead of sampled scores: the first
// half of the coverage counts without a score matrix (coverage_kernel.hpp)
__global__ void __launch_bounds__(FEED_G * WAVE, GARLIC_FEED_WG_PER_CU)
lod_bits_kernel(FeedArgs p)
{
    lod_feed_body<true>(p);
}

} // namespace garlic
