// ROH coverage counts straight from the genotypes: the window scores of calcLOD (src/garlic-roh.cpp:18-132) and the
// inWin[] loop of assembleROHWindows (src/garlic-roh.cpp:446-454) without a score matrix.
//
//   inWin[l] = #{ windows w in (l - W, l] of this individual with score >= cutoff }
//
// GARLIC's final pass computes every window score of the chosen size (8 B per window), keeps them, and counts.  Here
// the scores never exist in memory: lod_bits_kernel (feed_kernel.hpp: every wave a chain of its own, lane = individual)
// leaves ONE BIT per window and individual, and the count is a sliding sum over the last W bits, a pass with no chain
// in it at all (cov_counts_from_bits_kernel below): 2 B per window leave the chip instead of 8 B out + 8 B in again +
// 2 B out (garlic_lod_windows + garlic_roh_coverage), and no score matrix (100 GB at 10M SNPs x 1250 individuals) has
// to be resident.  (Rounds 2-3 also kept two one-kernel forms -- chain, compare and sliding count per lane -- that
// measured 1.9 x slower, 18 instructions per window on the run's critical path; they are in the history, DESIGN.md
// section 3 has the numbers.)
// Windows without a score never qualify here; the reference compares MISSING with the cutoff too, so the host only
// takes this path for cutoff > -9999 (and finite terms, and no -9999.0 sums: as for the thinned feed).
#pragma once
#include <type_traits>
#include "feed_kernel.hpp"
#include "cov_counts.hpp"

namespace garlic {

constexpr int COVF_MAX_W = 1024;

// ---- two kernels: bits, then counts.  lod_bits_kernel (feed_kernel.hpp) is the thinned feed's hand-scheduled chain
// with a compare and an add-with-carry per window instead of the sampled stores: it leaves ONE BIT per window and
// individual (7.1 instructions per window on the run's critical path); the counts are then a pass with no chain in it:
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
