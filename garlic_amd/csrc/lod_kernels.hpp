// Hand-written gfx950 (CDNA4, wave64) kernels for GARLIC's Phase-I window LOD scores.
//
// What the reference computes (src/garlic-roh.cpp:18-132, restated in SURVEY.md 8(a')):
//   * a pair of neighbouring SNPs (k-1,k) "breaks" when it is wider than MAX_GAP or touches the
//     centromere (garlic-roh.cpp:60-61,82-83,109-110); SNPs between two breaks form a *segment*;
//   * inside a segment [p,q] the windows starting at p .. q-W+1 are scored, every other window
//     of the chromosome is MISSING (-9999);
//   * per individual the first window of a segment is the left-to-right sum of its W per-SNP
//     terms, every following window is  (previous - leaving term) + entering term  -- two
//     separately rounded FP64 operations (garlic-roh.cpp:92-100).  That rolling sum is order
//     dependent, so bit-identity forces a sequential replay per (individual, segment): the SNP
//     axis is NOT scanned in parallel.  Parallelism = individuals x segments.
//
// Mapping onto the machine:
//   lane  = one individual, wavefront = 64 consecutive individuals of one segment (one workgroup
//           = one wave, so waves never wait on each other);
//   HBM   -> 2-bit genotypes, 16 SNPs per 32-bit word, individual-minor: a wave reads one
//           256-byte row per 16 SNPs (coalesced), twice (entering and leaving SNP streams);
//   LDS   <- the per-SNP term table {lod(g=0), lod(1), lod(2), 0.0} (host libm log10, see
//           garlic_hip.hip) for the 32 SNPs of the current tile, for both streams; a lane picks
//           its term with one ds_read_b64 at  row*32 + genotype*8;
//   LDS   <- a 64 x 32 transpose tile: lanes write their score column-wise, the wave reads it back
//           row-wise so every global store instruction writes 4 individuals x 256 contiguous
//           bytes of the individual-major output (WinData layout, src/garlic-data.h:83);
//   no MFMA: there is no contraction here, only a dependent add chain and byte movement.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "chain_loop_gfx950.inc"

namespace garlic {

constexpr int WAVE = 64;
constexpr int TILE = 32;          // window starts per tile (= 256 B of one output row)
constexpr int TPITCH = 34;        // doubles per LDS tile row: 272 B keeps ds_read_b128 16-B aligned
constexpr int GOFF = 32;          // pad SNP rows in front of the packed genotypes / term table
constexpr int GPAD_CHAIN = 32 * GARLIC_CHAIN_NSLOT + 48 * GARLIC_CHAIN_CHROWS + 64;
constexpr int GPAD_BACK = GPAD_CHAIN > 4160 ? GPAD_CHAIN : 4160;   // also: wLOD tiles read W + 63 SNPs past a chromosome end (W <= 4096) // pad SNP rows behind: the input rings run NSLOT tiles (+ one genotype chunk) ahead
constexpr double MISSING_D = -9999.0;

struct ChrDev {
    int64_t loc_base;   // global locus index of the chromosome's first SNP
    int64_t out_base;   // offset (doubles) of the chromosome block in the output
    int64_t out_pitch;  // row pitch (doubles)
    int32_t nloci;
    int32_t fast;       // layout allows the hand-scheduled loop (even pitch, 64-row padding, 32-bit row offsets)
};

struct ChainItem {   // one wavefront of work: a run of valid windows x 64 individuals
    int32_t chr;     // -1: padding item
    int32_t a;       // first valid window start (chromosome-local)
    int32_t b;       // last valid window start
    int32_t ind0;    // first individual (relative to the call's ind_begin)
};

struct FillItem {    // a stretch of MISSING windows [lo, hi) of one chromosome
    int32_t chr;
    int32_t lo;
    int32_t hi;
    int32_t pad;
};

// ------------------------------------------------------------------------------------------
// Genotype packing: int16 [locus][ind] (HapData::data, src/garlic-data.h:35) -> 2-bit codes.
// code 0/1/2 = genotype, 3 = anything else (missing, -9): lod() returns log10(1/1) = +0.0 for
// those (garlic-roh.cpp:379-383), which is entry 3 of the term table.
// One thread = one (word row, individual); consecutive lanes = consecutive individuals, so both
// the 16 strided reads and the write are coalesced.
//
// Device layout of the packed panel: [64-individual block][word row][64 individuals].  One chain
// item (SNP run x 64 individuals) then reads ONE sequential stream, 256 B per 16 SNPs, which the
// chain kernel fetches in 1 KB requests -- with the individual-minor layout of the whole panel
// the same bytes are 256 B pieces 4*nind bytes apart, and those scattered HBM reads, interleaved
// with the score write stream, cost 15 % of the kernel.
__device__ __host__ __forceinline__ int64_t packed_index(int64_t w, int64_t col, int64_t nwordrows)
{
    return (((col >> 6) * nwordrows + w) << 6) + (col & 63);
}

__global__ void pack_genotypes_kernel(const int16_t *__restrict__ geno, int64_t ld,
                                      int64_t locus_begin, int64_t locus_count, int32_t nind,
                                      int64_t nind_pad, int64_t nwordrows,
                                      uint32_t *__restrict__ packed, int64_t word_lo,
                                      int64_t word_hi)
{
    int64_t ind = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t w = word_lo + blockIdx.y;
    if (ind >= nind_pad || w >= word_hi) return;
    uint32_t *dst = packed + packed_index(w, ind, nwordrows);
    // SNPs of this word that the caller did not supply keep their previous bits
    uint32_t word = *dst;
    if (ind >= nind) { *dst = 0xFFFFFFFFu; return; }
    int64_t g0 = w * 16 - GOFF; // global locus of bit position 0
#pragma unroll
    for (int q = 0; q < 16; q++) {
        int64_t l = g0 + q - locus_begin;
        if (l >= 0 && l < locus_count) {
            int v = geno[l * ld + ind];
            uint32_t code = (v == 0) ? 0u : (v == 1) ? 1u : (v == 2) ? 2u : 3u;
            word = (word & ~(3u << (2 * q))) | (code << (2 * q));
        }
    }
    *dst = word;
}

// The same from SNP-major 2-bit rows (4 genotypes per byte, code 3 = missing; genotype cache / bed-like
// readers): rows[(l - locus_begin) * row_bytes + j / 4] >> 2 * (j % 4), j = ind_offset + shard-local
// individual.  An eighth of the bytes of the int16 rows on the host and over PCIe.
__global__ void pack_genotypes_2bit_kernel(const uint8_t *__restrict__ rows, int64_t row_bytes,
                                           int64_t ind_offset, int64_t locus_begin, int64_t locus_count,
                                           int32_t nind, int64_t nind_pad, int64_t nwordrows,
                                           uint32_t *__restrict__ packed, int64_t word_lo, int64_t word_hi)
{
    int64_t ind = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t w = word_lo + blockIdx.y;
    if (ind >= nind_pad || w >= word_hi) return;
    uint32_t *dst = packed + packed_index(w, ind, nwordrows);
    uint32_t word = *dst;
    if (ind >= nind) { *dst = 0xFFFFFFFFu; return; }
    const int64_t j = ind_offset + ind;
    const int sh = 2 * (int)(j & 3);
    int64_t g0 = w * 16 - GOFF;
#pragma unroll
    for (int q = 0; q < 16; q++) {
        int64_t l = g0 + q - locus_begin;
        if (l >= 0 && l < locus_count) {
            const uint32_t code = (rows[l * row_bytes + (j >> 2)] >> sh) & 3u;
            word = (word & ~(3u << (2 * q))) | (code << (2 * q));
        }
    }
    *dst = word;
}

__global__ void fill_u32_kernel(uint32_t *p, int64_t n, uint32_t v)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

// ------------------------------------------------------------------------------------------
// Segment boundaries: integer work, done with wavefront scans.
// boundary(G) = G is the first SNP of its chromosome, or the pair (G-1,G) breaks a window.
__device__ __forceinline__ int in_gap(int qs, int qe, int ts, int te)
{   // garlic-roh.cpp:11-16
    return (ts <= qs && te >= qs) || (ts <= qe && te >= qe) || (ts >= qs && te <= qe);
}

__device__ __forceinline__ int find_chr(const int64_t *chr_off, int nchr, int64_t G)
{   // largest c with chr_off[c] <= G
    int lo = 0, hi = nchr - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (chr_off[mid] <= G) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ int boundary_flag(const int32_t *pos, const int64_t *chr_off,
                                             const int32_t *cs, const int32_t *ce, int nchr,
                                             int64_t nloci, int32_t max_gap, int64_t G)
{
    if (G >= nloci) return 0;
    int c = find_chr(chr_off, nchr, G);
    if (G == chr_off[c]) return 1;
    int p0 = pos[G - 1], p1 = pos[G];
    return (p1 - p0 > max_gap) || in_gap(p0, p1, cs[c], ce[c]);
}

// inclusive prefix sum across the 64 lanes of a wavefront (__shfl_up ladder, no LDS)
__device__ __forceinline__ int wave_inclusive_scan(int v)
{
    int lane = threadIdx.x & (WAVE - 1);
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        int n = __shfl_up(v, d, WAVE);
        if (lane >= d) v += n;
    }
    return v;
}

constexpr int SEG_BLOCK = 256;
constexpr int SEG_ITEMS = 8; // SNPs per thread -> 2048 SNPs per workgroup

__global__ void __launch_bounds__(SEG_BLOCK)
seg_count_kernel(const int32_t *pos, const int64_t *chr_off, const int32_t *cs, const int32_t *ce,
                 int nchr, int64_t nloci, int32_t max_gap, int32_t *block_counts)
{
    __shared__ int wsum[SEG_BLOCK / WAVE];
    int64_t base = (int64_t)blockIdx.x * SEG_BLOCK * SEG_ITEMS;
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < SEG_ITEMS; k++)
        cnt += boundary_flag(pos, chr_off, cs, ce, nchr, nloci, max_gap,
                             base + (int64_t)k * SEG_BLOCK + threadIdx.x);
    int inc = wave_inclusive_scan(cnt);
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < SEG_BLOCK / WAVE; w++) t += wsum[w];
        block_counts[blockIdx.x] = t;
    }
}

// single wavefront: exclusive scan of the per-workgroup counts (nblocks is small)
__global__ void __launch_bounds__(WAVE)
seg_scan_kernel(const int32_t *block_counts, int nblocks, int32_t *block_offsets, int32_t *total)
{
    int carry = 0;
    for (int base = 0; base < nblocks; base += WAVE) {
        int i = base + threadIdx.x;
        int v = (i < nblocks) ? block_counts[i] : 0;
        int inc = wave_inclusive_scan(v);
        if (i < nblocks) block_offsets[i] = carry + inc - v;
        carry += __shfl(inc, WAVE - 1, WAVE);
    }
    if (threadIdx.x == 0) *total = carry;
}

// ordered compaction: boundaries[] receives the global loci of all segment starts, ascending
__global__ void __launch_bounds__(SEG_BLOCK)
seg_compact_kernel(const int32_t *pos, const int64_t *chr_off, const int32_t *cs,
                   const int32_t *ce, int nchr, int64_t nloci, int32_t max_gap,
                   const int32_t *block_offsets, int64_t *boundaries)
{
    __shared__ int wsum[SEG_BLOCK / WAVE];
    __shared__ int running;
    if (threadIdx.x == 0) running = block_offsets[blockIdx.x];
    int64_t base = (int64_t)blockIdx.x * SEG_BLOCK * SEG_ITEMS;
    __syncthreads();
    for (int k = 0; k < SEG_ITEMS; k++) {
        int64_t G = base + (int64_t)k * SEG_BLOCK + threadIdx.x;
        int f = boundary_flag(pos, chr_off, cs, ce, nchr, nloci, max_gap, G);
        int inc = wave_inclusive_scan(f);
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); w++) woff += wsum[w];
        int start = running;
        if (f) boundaries[start + woff + inc - 1] = G;
        __syncthreads();
        if (threadIdx.x == SEG_BLOCK - 1) running = start + woff + inc;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// MISSING stretches (the reference pre-fills WinData with -9999, src/garlic-data.cpp:1633; here
// every output element is written exactly once, so no memset pass over the whole output).
// grid.x = fill item, grid.y = group of rows; lanes run along the SNP axis (contiguous bytes).
constexpr int FILL_ROWS = 16;
__global__ void __launch_bounds__(256)
fill_missing_kernel(const FillItem *items, const ChrDev *chrs, int32_t nind, double *out)
{
    FillItem it = items[blockIdx.x];
    ChrDev c = chrs[it.chr];
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int row0 = blockIdx.y * FILL_ROWS;
    for (int r = row0 + wave; r < row0 + FILL_ROWS && r < nind; r += 4) {
        double *row = out + c.out_base + (int64_t)r * c.out_pitch;
        for (int l = it.lo + lane; l < it.hi; l += WAVE) row[l] = MISSING_D;
    }
}

// ------------------------------------------------------------------------------------------
// The chain kernel.
struct ChainArgs {
    const uint32_t *packed;   // [nind_pad/64][nwordrows][64], word row w holds global loci 16w-GOFF ..
    const double *tab;        // [GOFF + nloci + pad][4] per-SNP terms
    const ChainItem *items;
    const ChrDev *chrs;
    double *out;
    int64_t nind_pad;
    int64_t nwordrows;
    int32_t ind_begin;        // first individual of this call inside the panel shard
    int32_t ind_count;        // rows in the output
    int32_t winsize;
    int32_t n_items;          // work-list length
    int32_t *next_item;       // [0] the persistent workgroups' queue head, [1] workgroups that have left; both zero
                              // at launch (the last workgroup to leave resets them)
    int64_t *trace;           // optional (GARLIC_TRACE): per item {worker, t_begin, t_asm, t_end} in 100 MHz ticks
};

// LDS map (bytes).  One workgroup = 4 waves (CHAIN, POST, PRE, COMB roles) working on one item; the hand-scheduled loop
// (chain_loop_gfx950.inc, see tools/gen_chain_asm.py for the full map) owns everything from 4096
// on; the compiler-generated "generic" tile path used for a run's first and last tiles keeps its
// inputs in the slot at 0 and transposes through TILE buffer 0.
//   slot: for each of the two SNP streams (entering / leaving the window) the genotype words +1
//   and +2 of every lane (word +0 is carried in a register) and the 32 term rows
//   {lod(0),lod(1),lod(2),0.0} of the tile's SNPs.
constexpr uint32_t SLOT_BYTES = 3072;
constexpr uint32_t SL_LW1 = 0, SL_LW2 = 256, SL_TW1 = 512, SL_TW2 = 768;
constexpr uint32_t SL_LTAB = 1024, SL_TTAB = 2048;
constexpr uint32_t LDS_ITEM = 3072;                              // broadcast word for the item index
constexpr uint32_t LDS_TILE = GARLIC_CHAIN_LDS_TILE0;                             // TILE buffer 0 of the asm loop
constexpr uint32_t LDS_BYTES = GARLIC_CHAIN_LDS_TOTAL;
constexpr int CHAIN_THREADS = 256;   // 4 waves: CHAIN, POST, PRE, COMB

// Wave-uniform stream state (lives in SGPRs).
struct Streams {
    int64_t lead_w, trail_w;               // word row of the current tile's first SNP
    const double *lead_tab, *trail_tab;    // term row of the current tile's first SNP
    int sh_lead, sh_trail;                 // bit offset of the tile's first SNP in its word
};

__device__ __forceinline__ void advance(Streams &st)
{   // next tile = two genotype words and 32 term rows further along both streams
    st.lead_w += 2; st.trail_w += 2;
    st.lead_tab += 4 * TILE; st.trail_tab += 4 * TILE;
}

// same content through registers (compiler-managed waits): head / leftover tiles
// gcol = this lane's column of the packed panel (word row w at gcol[w * 64])
__device__ __forceinline__ void fill_regs(unsigned char *smem, const Streams &st,
                                          const uint32_t *gcol, int lane)
{
    const uint32_t l1 = gcol[(st.lead_w + 1) * WAVE], l2 = gcol[(st.lead_w + 2) * WAVE];
    const uint32_t t1 = gcol[(st.trail_w + 1) * WAVE], t2 = gcol[(st.trail_w + 2) * WAVE];
    const double2 tl = reinterpret_cast<const double2 *>(st.lead_tab)[lane];
    const double2 tt = reinterpret_cast<const double2 *>(st.trail_tab)[lane];
    *reinterpret_cast<uint32_t *>(smem + SL_LW1 + lane * 4) = l1;
    *reinterpret_cast<uint32_t *>(smem + SL_LW2 + lane * 4) = l2;
    *reinterpret_cast<uint32_t *>(smem + SL_TW1 + lane * 4) = t1;
    *reinterpret_cast<uint32_t *>(smem + SL_TW2 + lane * 4) = t2;
    *reinterpret_cast<double2 *>(smem + SL_LTAB + lane * 16) = tl;
    *reinterpret_cast<double2 *>(smem + SL_TTAB + lane * 16) = tt;
}

// Genotype bits of a tile's 32 steps, both streams (16 SNPs per word).
struct TileBits {
    uint32_t lead_lo, lead_hi, trail_lo, trail_hi;
};

// lc/tc: genotype word +0 of each stream (carried); funnel-shift the 3 words of a stream to the
// tile's first SNP.
template <int SLOT>
__device__ __forceinline__ TileBits tile_consume(const unsigned char *smem, uint32_t &lc,
                                                 uint32_t &tc, const Streams &st, int lane)
{
    const unsigned char *slot = smem + SLOT * SLOT_BYTES;
    const uint32_t l1 = *reinterpret_cast<const uint32_t *>(slot + SL_LW1 + lane * 4);
    const uint32_t l2 = *reinterpret_cast<const uint32_t *>(slot + SL_LW2 + lane * 4);
    const uint32_t t1 = *reinterpret_cast<const uint32_t *>(slot + SL_TW1 + lane * 4);
    const uint32_t t2 = *reinterpret_cast<const uint32_t *>(slot + SL_TW2 + lane * 4);
    TileBits tb;
    tb.lead_lo = __builtin_amdgcn_alignbit(l1, lc, st.sh_lead);
    tb.lead_hi = __builtin_amdgcn_alignbit(l2, l1, st.sh_lead);
    tb.trail_lo = __builtin_amdgcn_alignbit(t1, tc, st.sh_trail);
    tb.trail_hi = __builtin_amdgcn_alignbit(t2, t1, st.sh_trail);
    lc = l2;
    tc = t2;
    return tb;
}

// The 32 dependent steps of one tile.  EDGE: tile straddles the run's first or last window.
template <bool EDGE, int SLOT>
__device__ __forceinline__ void tile_steps(const unsigned char *smem, unsigned char *tile,
                                           double &acc, const TileBits &tb, int s0, int a, int b,
                                           int lane)
{
    const unsigned char *slot = smem + SLOT * SLOT_BYTES;
    const uint32_t tile_lane = (uint32_t)lane * (TPITCH * 8);
#pragma unroll
    for (int j = 0; j < TILE; j++) {
        const uint32_t lw = (j < 16) ? tb.lead_lo : tb.lead_hi;
        const uint32_t tw = (j < 16) ? tb.trail_lo : tb.trail_hi;
        const uint32_t g1 = (lw >> (2 * (j & 15))) & 3u;
        const uint32_t g0 = (tw >> (2 * (j & 15))) & 3u;
        double t_in = *reinterpret_cast<const double *>(slot + SL_LTAB + j * 32 + g1 * 8);
        double t_out = *reinterpret_cast<const double *>(slot + SL_TTAB + j * 32 + g0 * 8);
        if (EDGE) {
            const int s = s0 + j;
            // the first window of a run is a plain sum (no leaving term); steps outside [a,b]
            // leave the accumulator untouched (x - 0.0 + 0.0 == x for every x that can occur:
            // the accumulator starts at +0.0 and can never become -0.0)
            if (!(s > a && s <= b)) t_out = 0.0;
            if (!(s >= a && s <= b)) t_in = 0.0;
        }
        acc = (acc - t_out) + t_in; // two roundings, as garlic-roh.cpp:98-100
        *reinterpret_cast<double *>(tile + tile_lane + j * 8) = acc;
    }
}

// Transposed write-out of a tile: store q covers individuals 4q..4q+3, 16 lanes x 16 B = 256
// contiguous bytes of each row.  Every store is unconditional (the hand-counted vmcnt needs a
// fixed number of stores per tile): a 4-row group wholly past the shard's last individual
// re-stores group 0, a lane past the last row of a partial group re-stores that group's last
// valid row -- same address, same data as the lane that owns it.
// Address = wave-uniform row-group base (SGPR pair) + 32-bit lane offset.
template <bool EDGE, bool ALIGNED16>
__device__ __forceinline__ void tile_store(const unsigned char *tile, int s0, int a, int b, int lane,
                                           int rows_valid, double *out_tile, int64_t pitch)
{
    const int rsub = lane >> 4, csub = lane & 15;
#pragma unroll
    for (int q = 0; q < WAVE / 4; q++) {
        const int qe = (4 * q < rows_valid) ? q : 0;                 // uniform
        const int re = min(rsub, rows_valid - 1 - 4 * qe);           // per lane, 0..3
        const double2 v = *reinterpret_cast<const double2 *>(
            tile + (uint32_t)(4 * qe + re) * (TPITCH * 8) + csub * 16);
        char *base = reinterpret_cast<char *>(out_tile + (int64_t)(4 * qe) * pitch);
        const uint32_t off = (uint32_t)re * (uint32_t)(pitch * 8) + (uint32_t)csub * 16u;
        double *dst = reinterpret_cast<double *>(base + off);
        if (EDGE) {
            const int s = s0 + 2 * csub;
            if (s >= a && s <= b) dst[0] = v.x;
            if (s + 1 >= a && s + 1 <= b) dst[1] = v.y;
        } else if (!ALIGNED16) {
            dst[0] = v.x;
            dst[1] = v.y;
        } else {
            *reinterpret_cast<double2 *>(dst) = v;
        }
    }
}

__global__ void __launch_bounds__(256) fill_value_kernel(double *dst, int64_t n, double value)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = value;
}

template <bool ALIGNED16>
__global__ void __launch_bounds__(CHAIN_THREADS)
lod_chain_kernel(ChainArgs p)
{
    // one LDS object at offset 0: the hand-scheduled loop addresses it with absolute offsets
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    unsigned char *tile = smem + LDS_TILE;
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Persistent workgroup: pulls (run, 64-individual block) items, longest runs
    // first, from one device-wide counter.  Wave 0 owns the accumulator: it sums the run's first
    // window and runs the first / last (partial) tiles through the compiler-generated path; all
    // full tiles go through the 4-role hand-scheduled loop.
    for (;;) {
    if (threadIdx.x == 0)
        *reinterpret_cast<int *>(smem + LDS_ITEM) = atomicAdd(p.next_item, 1);
    __syncthreads();
    const int item_idx =
        __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int *>(smem + LDS_ITEM));   // (barriers on both sides)
    __syncthreads();
    if (item_idx >= p.n_items) {
        // the last workgroup to leave puts the queue head (and this exit count) back to zero for
        // the next launch: one operation less between two passes
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
        p.trace[4 * item_idx + 0] = blockIdx.x;
        p.trace[4 * item_idx + 1] = wall_clock64();
    }
    const ChainItem it = p.items[item_idx];
    const ChrDev c = p.chrs[it.chr];
    const int W = p.winsize;
    const int a = it.a, b = it.b;
    const int rows_valid = min(WAVE, p.ind_count - it.ind0);
    // lanes past the shard read the padded columns (code 3 -> term 0.0)
    const int64_t col0 = (int64_t)p.ind_begin + it.ind0;
    const uint32_t *gcol = p.packed + packed_index(0, col0 + lane, p.nwordrows);
    const int64_t Gbase = c.loc_base + GOFF; // global (padded) index of chromosome-local locus 0

    // ---- first window of the run: sum of W terms left to right (garlic-roh.cpp:57-71); the
    //      first W-1 of them here, the W-th enters in the first tile below.
    double acc = 0.0;
    if (wave == 0) {
        // 64 SNPs per round, every load of a round issued before the first add: 5 genotype words
        // (funnel-shifted to the round's first SNP), then 64 independent term gathers, then the
        // ordered adds -- two memory round trips per 64 terms instead of two per 16
        int l = a;
        const int lend = a + W - 1;
        while (l < lend) {
            const int64_t G = Gbase + l;
            const int sh = 2 * (int)(G & 15);
            const uint32_t *wp = gcol + (G >> 4) * WAVE;
            uint32_t wd[5];
#pragma unroll
            for (int q = 0; q < 5; q++) wd[q] = wp[q * WAVE];
            uint32_t al[4];
#pragma unroll
            for (int q = 0; q < 4; q++) al[q] = __builtin_amdgcn_alignbit(wd[q + 1], wd[q], sh);
            const int n = min(64, lend - l);
            double t[64];
#pragma unroll
            for (int q = 0; q < 64; q++) {
                const uint32_t g = (al[q >> 4] >> (2 * (q & 15))) & 3u;
                t[q] = p.tab[(G + min(q, n - 1)) * 4 + ((q < n) ? g : 3u)];
            }
#pragma unroll
            for (int q = 0; q < 64; q++) acc += (q < n) ? t[q] : 0.0;
            l += n;
        }
    }

    // ---- tiles of 32 window starts, aligned to 32 inside the chromosome
    int s0 = a & ~(TILE - 1);
    Streams st;
    {
        // entering-SNP stream starts at local locus s0+W-1, leaving-SNP stream at s0-1
        const int64_t Glead = Gbase + s0 + W - 1;
        const int64_t Gtrail = Gbase + s0 - 1;
        st.sh_lead = 2 * (int)(Glead & 15);
        st.sh_trail = 2 * (int)(Gtrail & 15);
        st.lead_w = Glead >> 4;
        st.trail_w = Gtrail >> 4;
        st.lead_tab = p.tab + Glead * 4;
        st.trail_tab = p.tab + Gtrail * 4;
    }
    double *const out_row = p.out + c.out_base + (int64_t)it.ind0 * c.out_pitch;
    double *out_tile = out_row + s0;     // full output only
    const int64_t pitch = c.out_pitch;

    // phase 0: the head tile alone, masked variant (the run's first window has no leaving term,
    //          even when the run starts exactly on a tile boundary); then all full tiles in the
    //          hand-scheduled loop;
    // phase 1: whatever is left (the partial tail tile; everything when the layout is not
    //          eligible for the fast loop), masked variant.
    for (int phase = 0; phase < 2; phase++) {
        const int last = (phase == 0) ? s0 : b;
        while (s0 <= last && s0 <= b) {
            if (wave == 0) {
                uint32_t lc = gcol[st.lead_w * WAVE];
                uint32_t tc = gcol[st.trail_w * WAVE];
                fill_regs(smem, st, gcol, lane);
                const TileBits tb = tile_consume<0>(smem, lc, tc, st, lane);
                tile_steps<true, 0>(smem, tile, acc, tb, s0, a, b, lane);
                tile_store<true, ALIGNED16>(tile, s0, a, b, lane, rows_valid, out_tile, pitch);
            }
            advance(st);
            s0 += TILE;
            out_tile += TILE;
        }
        // The hand-scheduled loop needs 256 B-aligned output rows, a block-aligned first
        // individual (one sequential genotype stream) and a window that fits its genotype ring.
        if (phase == 1 || !ALIGNED16 || !c.fast || (col0 & 63) != 0) continue;
        if (st.lead_w - st.trail_w > GARLIC_CHAIN_MAX_DW) continue;

        int ntiles = (b + 1 - s0) / TILE;
        if (ntiles >= 2) {
            // genotype word +0 of the first full tile, both streams (PRE funnel-shifts the words)
            uint32_t lc = gcol[st.lead_w * WAVE];
            uint32_t tc = gcol[st.trail_w * WAVE];
            // genotype ring of the loop: word row w lives at ring row w % WROWS, fetched in
            // chunks of CHROWS rows (1 KB requests).  Initial fill = chunk of the leaving stream's first word
            // .. one chunk past the entering stream's row of tile NSLOT-1.
            const int64_t chunk0 = st.trail_w / GARLIC_CHAIN_CHROWS;
            const int64_t chunkI = (st.lead_w + 2 * GARLIC_CHAIN_NSLOT) / GARLIC_CHAIN_CHROWS + 1;
            const uint32_t *pchunk =
                p.packed + packed_index(GARLIC_CHAIN_CHROWS * chunk0, col0, p.nwordrows);
            const uint32_t wmask = GARLIC_CHAIN_WROWS - 1;
            if (p.trace && threadIdx.x == 0) p.trace[4 * item_idx + 2] = wall_clock64();
            // all waves enter together; the block starts by draining each wave's own memory
            // operations, one barrier after CHAIN has reset the LDS counters, counters from
            // then on (TILE buffer 0 used by the head tile above is this wave's own)
            const uint32_t a_nchunk0 = (uint32_t)(chunkI - chunk0 + 1) * (GARLIC_CHAIN_CHROWS / 4);
            const uint32_t a_roff0 = (uint32_t)((GARLIC_CHAIN_CHROWS * chunk0) & wmask) * 256u;
            const uint32_t a_laddr0 = (uint32_t)(st.lead_w & wmask) * 256u;
            const uint32_t a_taddr0 = (uint32_t)(st.trail_w & wmask) * 256u;
            if (wave == 0) {
                // PRE's spreading tables (tools/gen_chain_asm.py, post_expand): entry b = the four genotypes
                // of byte b, one per byte, times 16 (entering stream) / times 64 (leaving stream).  They live
                // in the generic path's slot, which the head tile above has just finished with; every wave
                // drains its LDS operations and meets at a barrier at the top of the loop.
                uint32_t tin[4], tout[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t bb = 4u * (uint32_t)lane + (uint32_t)q;
                    const uint32_t sp = (bb & 3u) | (((bb >> 2) & 3u) << 8) | (((bb >> 4) & 3u) << 16) | (((bb >> 6) & 3u) << 24);
                    tin[q] = sp << 4;
                    tout[q] = sp << 6;
                }
                *reinterpret_cast<uint4 *>(smem + GARLIC_CHAIN_SPREAD_IN + lane * 16) = make_uint4(tin[0], tin[1], tin[2], tin[3]);
                *reinterpret_cast<uint4 *>(smem + GARLIC_CHAIN_SPREAD_OUT + lane * 16) = make_uint4(tout[0], tout[1], tout[2], tout[3]);
            }
            {
                asm volatile(GARLIC_CHAIN_LOOP_ASM
                             : [acc] "+v"(acc)
                             : [wave] "s"(wave), [lane] "v"(lane), [lc] "v"(lc), [tc] "v"(tc),
                               [pchunk] "s"(pchunk), [nchunk0] "s"(a_nchunk0), [roff0] "s"(a_roff0),
                               [laddr0] "s"(a_laddr0), [taddr0] "s"(a_taddr0),
                               [pltab] "s"(st.lead_tab), [pttab] "s"(st.trail_tab), [out] "s"(out_tile),
                               [ntiles] "s"(ntiles), [shl] "s"(st.sh_lead), [sht] "s"(st.sh_trail),
                               [pitch8] "s"((uint32_t)(pitch * 8)), [rows] "s"(rows_valid)
                             : GARLIC_CHAIN_LOOP_CLOBBERS);
            }
            st.lead_w += 2 * ntiles;
            st.trail_w += 2 * ntiles;
            st.lead_tab += (int64_t)(4 * TILE) * ntiles;
            st.trail_tab += (int64_t)(4 * TILE) * ntiles;
            s0 += ntiles * TILE;
            out_tile += (int64_t)ntiles * TILE;
            __syncthreads(); // the tail tile reuses LDS regions the loop's roles were reading
        }
    }
    if (p.trace && threadIdx.x == 0) p.trace[4 * item_idx + 3] = wall_clock64();
    } // next item
}

} // namespace garlic
