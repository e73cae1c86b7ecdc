// TGLS chain (src/garlic-roh.cpp:68,91-95,117 with the terms already in the term matrix), persistent form.
//
// lod_chain_terms_kernel (variant_kernels.hpp) keeps every work item resident at once, one 2-wave workgroup
// each: all items share HBM equally, the kernel lasts as long as the longest run's single wave, and the
// leaving term of every window is read from memory a second time.  This kernel is built like the
// unweighted one instead:
//
//   * 256 persistent workgroups (one per CU) pull (run, 64-individual block) items longest first;
//   * every term row (512 B = 64 individuals x 8 B; consecutive SNPs of a block are consecutive rows, so an
//     item reads ONE sequential stream) is fetched ONCE, by LDS-DMA, into a ring of TG_RING rows in LDS;
//     the window's entering term (row s+W-1) and leaving term (row s-1) are both read from that ring --
//     16 B of HBM traffic per window: 8 B of terms in, 8 B of scores out;
//   * four wavefronts, four roles, one per SIMD, decoupled by counters in LDS (no barrier per tile):
//       LOAD0 / LOAD1  the even / odd 1-KB requests (2 rows each) of the stream, 8 requests per round,
//                      TG_DEPTH rounds in flight each (counted vmcnt waits), never further ahead than the
//                      ring has room for;
//       CHAIN          per window two ds_read_b64 and the dependent acc = (acc - t_out) + t_in, scores into one
//                      of two 64 x 32 transpose tiles; no vector-memory instruction in its loop;
//       POST           transposed write-out of finished tiles (4 rows x 256 B per store, non-temporal).
//   * windows wider than the ring can span (W > TG_RING - 96) take the two-stream form: LOAD0 streams the
//     entering rows and LOAD1 the leaving rows into the two halves of the ring (the leaving stream is then a
//     second read, served by the caches as before).
//
// 16.25 B per window against HBM ("TGLS, GL as doubles").  Arithmetic and masks are those of
// lod_chain_terms_kernel: bit-identical output.
#pragma once
#include "variant_kernels.hpp"

namespace garlic {

#ifndef GARLIC_TG_RING      // (timing experiments override the three: tools/exp/tgls_ring_abl.sh)
#define GARLIC_TG_RING 240
#endif
#ifndef GARLIC_TG_DEPTH
#define GARLIC_TG_DEPTH 3
#endif
#ifndef GARLIC_TG_TILE_ROWS
#define GARLIC_TG_TILE_ROWS WAVE
#endif
constexpr int TG_RING = GARLIC_TG_RING;   // ring rows (512 B each): 120 KB
constexpr int TG_GROUP = 8;             // LDS-DMA requests (1 KB = 2 rows) per loader round
constexpr int TG_DEPTH = GARLIC_TG_DEPTH; // rounds a loader keeps in flight
constexpr int TG_THREADS = 4 * WAVE;    // LOAD0, LOAD1, CHAIN, POST
constexpr int TG_SINGLE_MAX_W = TG_RING - 32 - 4 * TG_GROUP * 2;   // one stream: W + the tile + a few rounds must fit

struct TglsArgs {
    const double *terms;      // [blk][term_rows][64]
    int64_t term_rows;
    const ChainItem *items;
    const ChrDev *chrs;
    double *out;
    int32_t ind_begin, ind_count, winsize, n_items;
    int32_t *next_item;       // [0] queue head, [1] workgroups that have left (both zero at launch; reset by the last one)
    CovBits cov;              // coverage bits instead of scores (variant_kernels.hpp): one dword per lane and tile; NULL: scores
};

// flags in LDS: [0] tiles written by CHAIN, [1] tiles stored by POST, [2] / [3] requests landed (LOAD0 / LOAD1)
__device__ __forceinline__ int tg_flag_read(const int *flag)
{   // inline asm: invisible to the compiler's wait-count pass, which would drain the loader's LDS-DMA
    // requests (they write LDS) in front of every LDS read it knows about
    int v;
    const uint32_t a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) int *)flag;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ void tg_flag_write(int *flag, int v)
{
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int *)flag;
    asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(a), "v"(v) : "memory");
}

// one loader wave: requests `n_req` 1-KB pieces (2 rows) of a row stream starting at `src`, piece i -> ring
// slot pair (first_slot + stride_slots * i) % ring_rows (+ ring_base), throttled by the consumer:
// piece i may be issued once  row_of(i) + 2 <= 32 * tiles_done + ring_rows  (rows behind the leaving
// stream are free).  Publishes the number of its pieces that have landed.
//   SINGLE: loader w takes pieces w, w+2, ..  (piece_step = 2, row_of(i) = 2 * (2i + w))
//   DOUBLE: loader w streams its own rows      (piece_step = 1, row_of(i) = 2i)
__device__ __forceinline__ void tg_loader(const double *src, int n_req, int piece_step, int w, int ring_rows,
                                          uint32_t ring_lds, const int *tiles_done, int *landed, int lane)
{
    const uint32_t lane16 = (uint32_t)lane * 16u;
    const uint32_t ring_bytes = (uint32_t)ring_rows * 512u;
    // piece i of this loader: stream piece index g = piece_step * i + (piece_step == 2 ? w : 0); its ring
    // offset and source address advance by one step per piece (ring_bytes is a multiple of the step)
    const uint32_t step_bytes = (uint32_t)piece_step * 1024u;
    uint32_t slot_off = (piece_step == 2 ? (uint32_t)w : 0u) * 1024u;
    const double *gp = src + (piece_step == 2 ? w : 0) * 128;                  // 2 rows = 128 doubles per piece
    int issued = 0, landed_n = 0;
    int inflight = 0;                      // full rounds issued and not yet known to have landed
    while (issued < n_req) {
        const int n = min(TG_GROUP, n_req - issued);
        // room: the last row of this round must not overwrite rows the chain still needs
        const int g_last = piece_step * (issued + n - 1) + (piece_step == 2 ? w : 0);
        const int need_done = (2 * g_last + 2 - ring_rows + 31) / 32;          // tiles that must be finished
        if (need_done > 0 && tg_flag_read(tiles_done) < need_done) {
            // The chain may be waiting for rows that are still in flight here: land and publish everything
            // before blocking (otherwise the two wait for each other when the window fills most of the ring)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (landed_n != issued) tg_flag_write(landed, issued);
            landed_n = issued;
            inflight = 0;
            while (tg_flag_read(tiles_done) < need_done) __builtin_amdgcn_s_sleep(2);
        }
#pragma unroll
        for (int q = 0; q < TG_GROUP; q++) {
            if (q < n) {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                             :: "s"(ring_lds + slot_off), "v"(lane16), "s"(gp) : "memory");
                slot_off += step_bytes;
                if (slot_off >= ring_bytes) slot_off -= ring_bytes;
                gp += piece_step * 128;
            }
        }
        issued += n;
        if (n < TG_GROUP) break;           // a short last round: drained below
        // requests retire in issue order: with TG_DEPTH full rounds out, wait for the oldest one
        if (++inflight == TG_DEPTH) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TG_GROUP * (TG_DEPTH - 1)) : "memory");
            landed_n += TG_GROUP;
            tg_flag_write(landed, landed_n);
            inflight--;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tg_flag_write(landed, n_req);
}

__global__ void __launch_bounds__(TG_THREADS)
lod_chain_ring_kernel(TglsArgs p)
{
    __shared__ __attribute__((aligned(1024))) double ring[TG_RING * WAVE];
    __shared__ __attribute__((aligned(16))) double tiles[2][GARLIC_TG_TILE_ROWS * TPITCH];
    __shared__ int flags[8];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double *)ring;
    for (;;) {
        if (threadIdx.x == 0) {
            flags[4] = atomicAdd(p.next_item, 1);
            flags[0] = 0; flags[1] = 0; flags[2] = 0; flags[3] = 0;
        }
        __syncthreads();
        const int item_idx = __builtin_amdgcn_readfirstlane(flags[4]);
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
        const ChainItem it = p.items[item_idx];
        const ChrDev c = p.chrs[it.chr];
        const int W = p.winsize, a = it.a, b = it.b;
        const int first = a & ~(TILE - 1);
        const int ntiles = (b - first) / TILE + 1;
        const bool single = W <= TG_SINGLE_MAX_W;
        const int ring_rows = single ? TG_RING : TG_RING / 2;
        const int64_t col0 = (int64_t)p.ind_begin + it.ind0;            // block-aligned (host-checked)
        const int64_t Gbase = c.loc_base + GOFF;
        const double *blk = p.terms + ((col0 >> 6) * p.term_rows) * WAVE;   // the block's rows, 64 doubles each
        // row streams: leaving rows start at local locus first - 1, entering rows at first + W - 1
        const double *trail = blk + (Gbase + first - 1) * WAVE;
        const double *lead = blk + (Gbase + first + W - 1) * WAVE;

        if (wave < 2) {   // ---- loaders
            if (single) {
                // one stream, rows first-1 .. first + 32*ntiles - 1 + W - 1, an even count of them
                const int n_rows = TILE * ntiles + W;
                const int n_pieces = (n_rows + 1) / 2;
                const int mine = (n_pieces - wave + 1) / 2;              // pieces wave, wave + 2, ..
                tg_loader(trail, mine, 2, wave, ring_rows, ring_lds, &flags[0], &flags[2 + wave], lane);
            } else {
                const int n_pieces = TILE * ntiles / 2;
                tg_loader(wave == 0 ? lead : trail, n_pieces, 1, 0, ring_rows,
                          ring_lds + (wave == 0 ? 0u : (uint32_t)ring_rows * 512u), &flags[0], &flags[2 + wave], lane);
            }
        } else if (wave == 3 && p.cov.bits) {
            // coverage bits: the chain wave stores its own dword per tile, there are no tiles to write out
        } else if (wave == 3) {   // ---- write-out
            const int rows_valid = min(WAVE, p.ind_count - it.ind0);
            double *out_row0 = p.out + c.out_base + (int64_t)it.ind0 * c.out_pitch;
            for (int k = 0; k < ntiles; k++) {
                while (LDS_FLAG_GET(flags[0]) <= k) __builtin_amdgcn_s_sleep(1);
                lds_acquire();
                variant_store(tiles[k & 1], first + k * TILE, a, b, lane, rows_valid, out_row0 + first + k * TILE, c.out_pitch);
                lds_release();
                if (lane == 0) LDS_FLAG_SET(flags[1], k + 1);
            }
        } else {   // ---- chain
            // first window of the run: its first W-1 terms, left to right (garlic-roh.cpp:57-71); the W-th
            // enters in the first tile.  Straight from memory, 32 loads in flight (once per item).
            const double *tcol = blk + lane;
            double acc = 0.0;
            for (int l0 = a; l0 < a + W - 1; l0 += 32) {
                double t[32];
#pragma unroll
                for (int q = 0; q < 32; q++) t[q] = tcol[(Gbase + min(l0 + q, a + W - 2)) * WAVE];
#pragma unroll
                for (int q = 0; q < 32; q++) acc += (l0 + q < a + W - 1) ? t[q] : 0.0;
            }
            // ring slot of the leaving / entering row of window first + i:
            //   single: (i) % R and (i + W) % R of one ring;  double: (i) % R2 in the upper / lower half
            int so = 0, si = single ? W % ring_rows : 0;
            const int base_out = single ? 0 : ring_rows * WAVE;
            // coverage bits: the lane's row of the bit matrix (looked up once per item: inside the tile loop the load
            // sat on the chain's critical path, a trip to memory per tile)
            uint32_t *bits_row = nullptr;
            if (p.cov.bits) {
                const ChrDev bc = p.cov.bchrs[it.chr];
                bits_row = p.cov.bits + bc.out_base + (int64_t)(it.ind0 + lane) * bc.out_pitch;
            }
            for (int k = 0; k < ntiles; k++) {
                // inputs: every row this tile reads has landed
#ifndef GARLIC_TG_ABL_NOWAIT     // (timing experiment when defined: the chain never waits for its rows)
                if (single) {
                    const int pieces = (TILE * (k + 1) + W + 1) / 2;            // stream pieces 0 .. pieces-1
                    const int need0 = (pieces + 1) / 2, need1 = pieces / 2;      // of loader 0 (even) / 1 (odd)
                    while (LDS_FLAG_GET(flags[2]) < need0 || LDS_FLAG_GET(flags[3]) < need1) __builtin_amdgcn_s_sleep(1);
                } else {
                    const int need = TILE * (k + 1) / 2;
                    while (LDS_FLAG_GET(flags[2]) < need || LDS_FLAG_GET(flags[3]) < need) __builtin_amdgcn_s_sleep(1);
                }
#endif
                const bool bits_mode = p.cov.bits != nullptr;
                if (!bits_mode)
                    while (LDS_FLAG_GET(flags[1]) + 2 <= k) __builtin_amdgcn_s_sleep(1);   // tile buffer k & 1 written out
                lds_acquire();
                double *tile = tiles[k & 1];
                const int s0 = first + k * TILE;
                const bool edge = (s0 <= a) || (s0 + TILE - 1 > b);
                double t_in[TILE], t_out[TILE];
                if (si + TILE <= ring_rows && so + TILE <= ring_rows) {
                    // neither stream wraps inside this tile (six tiles in seven): one address per stream, the rows at
                    // immediate offsets -- with a wrap test per row the reads cost ten instructions per window, more
                    // than the chain itself, and the chain wave is what paces the kernel when it leaves bits
                    const double *pi = ring + si * WAVE + lane, *po = ring + base_out + so * WAVE + lane;
#pragma unroll
                    for (int j = 0; j < TILE; j++) {
#ifdef GARLIC_TG_ABL_NOREADS        // (timing experiment: no ring reads)
                        t_in[j] = (double)(j + k);
                        t_out[j] = (double)(j - k);
#else
                        t_in[j] = pi[j * WAVE];
                        t_out[j] = po[j * WAVE];
#endif
                    }
                    si = (si + TILE == ring_rows) ? 0 : si + TILE;
                    so = (so + TILE == ring_rows) ? 0 : so + TILE;
                } else {
#pragma unroll
                    for (int j = 0; j < TILE; j++) {
                        t_in[j] = ring[si * WAVE + lane];
                        t_out[j] = ring[base_out + so * WAVE + lane];
                        si = (si + 1 == ring_rows) ? 0 : si + 1;
                        so = (so + 1 == ring_rows) ? 0 : so + 1;
                    }
                }
                if (bits_mode) {
                    // one bit per window -- score >= cutoff, NaN never -- 32 per lane and tile; a tile at a run's edge may
                    // hold another run's windows too when W < 32: those OR their bits into the zeroed matrix
                    uint32_t m = 0;
                    const double cutoff = p.cov.cutoff;
                    if (edge) {
#pragma unroll
                        for (int j = 0; j < TILE; j++) {
                            const int s = s0 + j;
                            const bool in = (s >= a && s <= b);
                            const double ti = in ? t_in[j] : 0.0;
                            const double to = (in && s > a) ? t_out[j] : 0.0;
                            acc = (acc - to) + ti;
                            m |= (in && acc >= cutoff) ? (1u << j) : 0u;
                        }
                    } else {        // (a loop of its own: with the edge tests in it every window carried a dozen more instructions)
#ifdef GARLIC_TG_ABL_NOCOMPUTE      // (timing experiment: the reads, no chain)
                        asm volatile("" :: "v"(t_in[0]), "v"(t_out[0]), "v"(t_in[TILE - 1]), "v"(t_out[TILE - 1]));
#else
#pragma unroll
                        for (int j = 0; j < TILE; j++) {
                            acc = (acc - t_out[j]) + t_in[j];
                            m |= (acc >= cutoff) ? (1u << j) : 0u;
                        }
#endif
                    }
                    if (it.ind0 + lane < p.ind_count) {
                        uint32_t *w = bits_row + (s0 >> 5);
                        if (edge) { if (m) atomicOr(w, m); }
                        else *w = m;
                    }
                } else if (edge) {
#pragma unroll
                    for (int j = 0; j < TILE; j++) {
                        const int s = s0 + j;
                        const bool in = (s >= a && s <= b);
                        const double ti = in ? t_in[j] : 0.0;
                        const double to = (in && s > a) ? t_out[j] : 0.0;
                        acc = (acc - to) + ti;
                        tile[lane * TPITCH + j] = acc;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < TILE; j++) {
                        acc = (acc - t_out[j]) + t_in[j];
                        tile[lane * TPITCH + j] = acc;
                    }
                }
                lds_release();
                if (lane == 0) LDS_FLAG_SET(flags[0], k + 1);
            }
        }
        __syncthreads();   // the item's rings, tiles and counters are free again
    }
}

} // namespace garlic
