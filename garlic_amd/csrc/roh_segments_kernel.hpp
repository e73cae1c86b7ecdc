// ROH segments on the device: the second half of assembleROHWindows (src/garlic-roh.cpp:456-533) from the one-bit-per-
// window rows the coverage kernels leave (coverage_kernel.hpp, feed_kernel.hpp, variant_kernels.hpp).
//
// The reference fills inWin[] (2 B per individual and SNP), then walks every row with a four-branch state machine.
// What that machine reports is, per individual and chromosome:
//   r[w]      = inWin[w] >= OVERLAP_FRAC * winsize (clamped to [1, winsize])            SNP w is "in ROH"
//   break[w]  = pos[w] - pos[w-1] > MAX_GAP || the pair straddles the centromere         (w >= 1; the windows' own rule)
//   segments  = the maximal stretches of consecutive r = 1 SNPs, cut in front of every SNP with break = 1,
//               kept when their SNP count reaches the same threshold -- except a stretch that BEGINS at the
//               chromosome's last SNP, which the machine opens and never closes (:456-468 take the first or second
//               branch there, the closing fourth one is an else-if behind them).
// A chromosome whose first SNP sits at position 0 (a 0-based map) is the one case where the machine's tests on winStart --
// a POSITION, "< 0" for no segment open, "> 0" for one open (:456, :493, :514) -- disagree: a stretch opened at SNP 0 is
// neither.  It cannot be closed by an uncovered SNP or by the chromosome's end, nothing can be opened while it lasts, and
// it ends at the first covered SNP behind a break (second branch), reported as SNPs 0 .. that SNP - 1 whatever lies
// between.  roh_wedge_kernel finds that SNP per individual (rows whose SNP 0 is in ROH), reports the one segment and
// tells roh_segments_from_mask_kernel to drop everything in front of it.  Negative positions are refused by the host.
// So nothing 2-byte-per-SNP has to exist: roh_mask_from_bits_kernel turns the window bits into r (the sliding count of
// cov_counts.hpp in registers, compared, 32 SNPs per thread = one dword), roh_segments_from_mask_kernel finds every
// segment's last SNP, walks back to its first and appends (individual, chromosome, first, last) to a list: a few MB
// instead of 25 GB of counts at 10M SNPs x 1250 individuals.  The list comes out unordered; the host sorts it into
// the reference's order (individual, chromosome, position).
#pragma once
#include "lod_kernels.hpp"
#include "cov_counts.hpp"
#include "../../include/garlic_hip.h"

namespace garlic {

// scores -> window bits, for the cases the bit-writing chains do not take (the same fallbacks as the counts):
// bit = score >= cutoff, as src/garlic-roh.cpp:449 (MISSING is compared like any score -- it only qualifies for a cutoff
// at or below -9999, where the counts stop at the chromosome's end as in garlic_roh_coverage --; NaN never qualifies)
__global__ void __launch_bounds__(256)
roh_bits_from_scores_kernel(const double *__restrict__ scores, const ChrDev *__restrict__ schrs, const ChrDev *__restrict__ bchrs,
                            const int32_t *__restrict__ word_base, int nchr, int W, double cutoff, uint32_t *__restrict__ bits)
{
    const int g = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (g >= word_base[nchr]) return;
    const int chr = cov_word_chr(word_base, nchr, g);
    const ChrDev sc = schrs[chr], bc = bchrs[chr];
    const int t = g - word_base[chr], row = blockIdx.y;
    const double *srow = scores + sc.out_base + (int64_t)row * sc.out_pitch;
    uint32_t m = 0;
    for (int j = 0; j < 32; j++) {
        const int w = 32 * t + j;
        if (w < sc.nloci && srow[w] >= cutoff) m |= 1u << j;
    }
    bits[bc.out_base + (int64_t)row * bc.out_pitch + t] = m;
}

// break[w] bits per chromosome word (the same for every individual) from the panel's run boundaries: global loci that
// begin a run of SNPs without a break; a chromosome's first SNP is a boundary but not a break
__global__ void __launch_bounds__(256)
roh_break_bits_kernel(const int64_t *__restrict__ boundaries, int n, const int64_t *__restrict__ chr_off,
                      const int32_t *__restrict__ word_base, int nchr, uint32_t *__restrict__ brk)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const int64_t G = boundaries[i];
    const int c = find_chr(chr_off, nchr, G);
    const int64_t w = G - chr_off[c];
    if (w > 0) atomicOr(brk + word_base[c] + (int)(w >> 5), 1u << (w & 31));
}

// r bits: thread per (individual, 32-SNP word); the count in front of the word by popcounts, then bit in / bit out
constexpr int ROH_ROWS = 8;      // individuals per workgroup: a launch of one workgroup per 256 words and row runs at the dispatcher's pace

__global__ void __launch_bounds__(256)
roh_mask_from_bits_kernel(const uint32_t *__restrict__ bits, const ChrDev *__restrict__ bchrs,
                          const int32_t *__restrict__ word_base, int nchr, int nind, int W, int thr, uint32_t *__restrict__ mask)
{
    const int g = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (g >= word_base[nchr]) return;
    const int chr = cov_word_chr(word_base, nchr, g);
    const ChrDev bc = bchrs[chr];
    const int t = g - word_base[chr], nwords = (bc.nloci + 31) >> 5;
    const int rel = 32 * t - W, dA = rel >> 5, r = rel & 31;          // bit 32 t - W sits in word dA at bit r (floor)
    const int left = bc.nloci - 32 * t;                                // SNPs of the chromosome in this word
    const int row_end = min(nind, ((int)blockIdx.y + 1) * ROH_ROWS);
    for (int row = (int)blockIdx.y * ROH_ROWS; row < row_end; row++) {
        const uint32_t *brow = bits + bc.out_base + (int64_t)row * bc.out_pitch;
        auto word = [&](int x) -> uint32_t { return (x >= 0 && x < nwords) ? brow[x] : 0u; };
        const uint32_t cur = word(t);
        const uint32_t wA = word(dA);
        const uint32_t F = __builtin_amdgcn_alignbit(word(dA + 1), wA, (uint32_t)r);      // bit j = window 32 t - W + j
        int cnt = (int)__popc(wA >> r);                               // windows 32 t - W .. 32 t - 1
        for (int x = dA + 1; x < t; x += 4) {                         // (four loads in flight)
            const uint32_t w0 = word(x), w1 = x + 1 < t ? word(x + 1) : 0u, w2 = x + 2 < t ? word(x + 2) : 0u,
                           w3 = x + 3 < t ? word(x + 3) : 0u;
            cnt += (int)(__popc(w0) + __popc(w1) + __popc(w2) + __popc(w3));
        }
        // inside the word the count moves between cnt - popc(F) and cnt + popc(cur): most words are decided by that alone
        // (SNPs far from any qualifying window, or deep inside a stretch of them); the others walk their 32 steps
        uint32_t m = 0;
        if (cnt - (int)__popc(F) >= thr) {      // (__popc returns unsigned)
            m = ~0u;
        } else if (cnt + (int)__popc(cur) >= thr) {
#pragma unroll
            for (int j = 0; j < 32; j++) {
                cnt += (int)((cur >> j) & 1u) - (int)((F >> j) & 1u);
                m |= (cnt >= thr) ? (1u << j) : 0u;
            }
        }
        if (left < 32) m &= (1u << left) - 1u;
        mask[bc.out_base + (int64_t)row * bc.out_pitch + t] = m;
    }
}

// first / last SNPs of the segments inside word t of a row: start = r & (!r[w-1] | break[w]), end = r & (!r[w+1] | break[w+1])
__device__ __forceinline__ uint32_t roh_start_bits(const uint32_t *mrow, const uint32_t *brk, int t)
{
    const uint32_t R = mrow[t], prev = t > 0 ? mrow[t - 1] >> 31 : 0u;
    return R & (~((R << 1) | prev) | brk[t]);
}

// chromosomes that start at position 0 (wedge_chr[0 .. n_wedge)): thread per (individual, such chromosome).
// w0[chr * nind + row] = -1: SNP 0 not in ROH, the row is an ordinary one; otherwise the first covered SNP behind a break
// (INT_MAX: none -- the row reports nothing on this chromosome)
constexpr int ROH_NO_BREAK = 0x7fffffff;
__global__ void __launch_bounds__(64)
roh_wedge_kernel(const uint32_t *__restrict__ mask, const ChrDev *__restrict__ bchrs, const uint32_t *__restrict__ brk,
                 const int32_t *__restrict__ word_base, const int32_t *__restrict__ wedge_chr, int n_wedge, int nind, double T,
                 int32_t *__restrict__ w0, garlic_roh_segment *__restrict__ segs, long long cap, unsigned long long *__restrict__ count)
{
    const int g = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (g >= n_wedge * nind) return;
    const int chr = wedge_chr[g / nind], row = g % nind;
    const ChrDev bc = bchrs[chr];
    const uint32_t *mrow = mask + bc.out_base + (int64_t)row * bc.out_pitch, *b = brk + word_base[chr];
    int first = -1;
    if (bc.nloci > 0 && (mrow[0] & 1u)) {
        first = ROH_NO_BREAK;
        const int nwords = (bc.nloci + 31) >> 5;
        for (int t = 0; t < nwords; t++) {
            const uint32_t hit = mrow[t] & b[t];        // (bit 0 of word 0 is never a break)
            if (hit) { first = 32 * t + __builtin_ctz(hit); break; }
        }
        if (first != ROH_NO_BREAK && (double)first >= T) {      // SNPs 0 .. first - 1: first of them
            const unsigned long long slot = atomicAdd(count, 1ull);
            if ((long long)slot < cap) segs[slot] = garlic_roh_segment{row, chr, 0, first - 1};
        }
    }
    w0[(int64_t)chr * nind + row] = first;
}

__global__ void __launch_bounds__(256)
roh_segments_from_mask_kernel(const uint32_t *__restrict__ mask, const ChrDev *__restrict__ bchrs,
                              const uint32_t *__restrict__ brk, const int32_t *__restrict__ word_base, int nchr, int nind,
                              double T, garlic_roh_segment *__restrict__ segs, long long cap, unsigned long long *__restrict__ count,
                              const int32_t *__restrict__ w0)
{
    const int g = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (g >= word_base[nchr]) return;
    const int chr = cov_word_chr(word_base, nchr, g);
    const ChrDev bc = bchrs[chr];
    const int t = g - word_base[chr], nwords = (bc.nloci + 31) >> 5;
    const uint32_t *b = brk + word_base[chr];
    const uint32_t bt = b[t], bnext = t + 1 < nwords ? (b[t + 1] & 1u) : 0u;
    const int row_end = min(nind, ((int)blockIdx.y + 1) * ROH_ROWS);
    for (int row = (int)blockIdx.y * ROH_ROWS; row < row_end; row++) {
        const uint32_t *mrow = mask + bc.out_base + (int64_t)row * bc.out_pitch;
        const uint32_t R = mrow[t];
        if (!R) continue;
        const int from = w0 ? w0[(int64_t)chr * nind + row] : -1;     // (w0: NULL unless a chromosome starts at position 0)
        const uint32_t next = t + 1 < nwords ? (mrow[t + 1] & 1u) : 0u;
        uint32_t end = R & (~((R >> 1) | (next << 31)) | ((bt >> 1) | (bnext << 31)));
        if (!end) continue;
        const uint32_t start_here = roh_start_bits(mrow, b, t);
        while (end) {
            const int e = __builtin_ctz(end);
            end &= end - 1;
            // the segment's first SNP: the highest start bit at or below e, in this word or in one before it
            int x = t;
            uint32_t s_bits = start_here & (e == 31 ? ~0u : ((2u << e) - 1u));
            while (!s_bits && x > 0) {     // (every word between is all ones: a stretch of r = 1 has a first SNP)
                x--;
                s_bits = roh_start_bits(mrow, b, x);
            }
            const int s = 32 * x + 31 - __builtin_clz(s_bits), stop = 32 * t + e;
            const int len = stop - s + 1;
            if ((double)len >= T && s != bc.nloci - 1 && s >= from) {     // (a break cuts at `from`: no segment straddles it)
                const unsigned long long slot = atomicAdd(count, 1ull);
                if ((long long)slot < cap) segs[slot] = garlic_roh_segment{row, chr, s, stop};
            }
        }
    }
}

} // namespace garlic
