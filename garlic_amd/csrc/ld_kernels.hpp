// LD weights of wLOD on the device: calcLDData / calcHR2LD / hr2 (src/garlic-data.cpp:330-583),
// the unphased homozygosity-correlation variant GARLIC uses unless --phased is given.
//
//   homFreq[l]  = #(genotype 0 or 2) / #(non-missing) over ALL individuals      (:656-676)
//   hr2(i, j)   = 0 unless 0 < homFreq < 1 at both SNPs; else with
//                   total = #(both non-missing), HAB = #(both non-missing and both homozygous)
//                 over the LD subsample:  HAB /= total;  H = HAB - HA*HB;
//                 min(1, H*H / (HA*(1-HA)*HB*(1-HB)))                               (:558-583)
//   LD[s][k]    = sum_{i = s .. s+W-1, in this order, from 0.0}  (i == s+k ? 1 : hr2(i, s+k))
//                 for s <= nloci_c - W, 0 elsewhere                                  (:474-527)
//
// --phased (calcR2LD / r2, :426-535, 585-617) differs only in the pair statistic: with p = FreqData::freq,
//   r2(i, j)    = 0 unless 0 < p < 1 at both SNPs; else with, over the LD subsample,
//                   total = 2 * #(both non-missing),
//                   x11   = 2 * #(2,2) + #(1,2) + #(2,1) + #(1,1 and firstCopy equal)
//                 x11 /= total;  D = x11 - pi*pj;  min(1, D*D / (pi*(1-pi)*pj*(1-pj)))
// -- the same expression as hr2 with (HA, HB, HAB, total) := (pi, pj, x11, total), so only the
// planes and the pair counts have a phased variant.
//
// Split so that individuals can be sharded over GPUs: everything that depends on genotypes is an
// INTEGER count (exact, order-free), summed over shards by the caller (RCCL all-reduce); the
// floating-point part is replicated and runs in the reference's operation order.
//
//   ld_planes_kernel   packed 2-bit genotypes -> per-SNP bit planes over individuals (ballots),
//                      per-SNP {homozygous, non-missing} counts of the shard
//   ld_pair_kernel     pair counts {total, HAB} for j - i = 1 .. W-1 (AND + popcount on the planes)
//   ld_hr2_kernel      counts -> hr2(i, i+d) and hr2(i+d, i)   (the formula is not symmetric in
//                      floating point: the denominator is multiplied in argument order)
//   ld_sum_kernel      the ordered window sums
#pragma once
#include "lod_kernels.hpp"
#include "tgls_math.hpp"
#include "variant_kernels.hpp"   // reciprocal_x86

namespace garlic {

// 0/0 (a SNP pair no sampled individual has both genotypes for; a SNP nobody is genotyped at) is the
// one invalid operation of this path.  x86 answers with its default NaN, sign bit set, and every
// later operation hands that NaN on unchanged; gfx950's default NaN has the sign clear.  The value
// is made x86's where it arises, so that LD weights, wLOD scores and --raw-lod text ("-nan") match.
__device__ __forceinline__ double x86_nan_if_nan(double v)
{
    return v != v ? f64_from_bits(X86_DEFAULT_NAN) : v;
}

// lane `lane` of `old` := the wave-uniform value v (v_writelane_b32)
__device__ __forceinline__ uint32_t ld_writelane(uint32_t v, int lane, uint32_t old)
{
    asm("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(v), "n"(lane));     // (lane: a constant after unrolling)
    return old;
}

// One workgroup per genotype word row (16 SNPs); wave w takes the 64-individual blocks w, w+4, ...
// planes: [blk][nloci] 64-bit masks, bit = individual of the block;  M = non-missing and in the LD
// subsample, H = M and homozygous.  counts: [nloci][2] = {homozygous, non-missing} over every
// individual of the shard (homFreq does not use the subsample, garlic-data.cpp:656).
// PHASED: planeH holds "genotype 2" instead of "homozygous", planeO "genotype 1" (both within the
// subsample); the counts stay {homozygous, non-missing}.
template <bool PHASED>
__global__ void __launch_bounds__(256)
ld_planes_kernel(const uint32_t *__restrict__ packed, int64_t nwordrows, int nblk,
                 const uint64_t *__restrict__ submask, int64_t nloci, uint64_t *__restrict__ planeM,
                 uint64_t *__restrict__ planeH, uint64_t *__restrict__ planeO, int32_t *__restrict__ counts)
{
    __shared__ int32_t red[4][16][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t w = blockIdx.x;
    const int64_t l0 = w * 16 - GOFF;           // unpadded global index of the word's first SNP
    int32_t hom = 0, tot = 0;                    // lane q < 16 owns SNP l0 + q
    // the wave's blocks eight at a time, their words requested together (one trip to memory per
    // eight blocks instead of one per block)
    for (int blk0 = wave; blk0 < nblk; blk0 += 32) {
        uint32_t words[8];
#pragma unroll
        for (int u = 0; u < 8; u++)
            words[u] = packed[((int64_t)min(blk0 + 4 * u, nblk - 1) * nwordrows + w) * WAVE + lane];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int blk = blk0 + 4 * u;
            if (blk >= nblk) break;
            const uint32_t word = words[u];
            const uint64_t sub = submask[blk];
            // the 16 SNPs' masks (wave-uniform ballots) go to lanes 0 .. 15 of four registers, then leave as ONE
            // 128-B store per plane (a store per SNP from its own lane: 32 single-lane stores per word, and the
            // kernel ran at the pace of those -- 13.5 ms at 10M SNPs x 1250)
            uint32_t mlo = 0, mhi = 0, hlo = 0, hhi = 0, olo = 0, ohi = 0, tlo = 0, thi = 0;
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const uint32_t code = (word >> (2 * q)) & 3u;
                const uint64_t m = __ballot(code != 3u);
                const uint64_t h = __ballot(code == 0u || code == 2u);
                mlo = ld_writelane((uint32_t)m, q, mlo);
                mhi = ld_writelane((uint32_t)(m >> 32), q, mhi);
                hlo = ld_writelane((uint32_t)h, q, hlo);
                hhi = ld_writelane((uint32_t)(h >> 32), q, hhi);
                if (PHASED) {
                    const uint64_t two = __ballot(code == 2u), one = __ballot(code == 1u);
                    tlo = ld_writelane((uint32_t)two, q, tlo);
                    thi = ld_writelane((uint32_t)(two >> 32), q, thi);
                    olo = ld_writelane((uint32_t)one, q, olo);
                    ohi = ld_writelane((uint32_t)(one >> 32), q, ohi);
                }
            }
            if (lane < 16) {
                const uint64_t m = ((uint64_t)mhi << 32) | mlo, h = ((uint64_t)hhi << 32) | hlo;
                hom += __popcll(h);
                tot += __popcll(m);
                const int64_t l = l0 + lane;
                if (l >= 0 && l < nloci) {
                    planeM[(int64_t)blk * nloci + l] = m & sub;
                    if (!PHASED) planeH[(int64_t)blk * nloci + l] = h & sub;
                    if (PHASED) {
                        planeH[(int64_t)blk * nloci + l] = (((uint64_t)thi << 32) | tlo) & sub;
                        planeO[(int64_t)blk * nloci + l] = (((uint64_t)ohi << 32) | olo) & sub;
                    }
                }
            }
        }
    }
    if (lane < 16) { red[wave][lane][0] = hom; red[wave][lane][1] = tot; }
    __syncthreads();
    if (threadIdx.x < 16) {
        const int64_t l = l0 + threadIdx.x;
        if (l >= 0 && l < nloci) {
            counts[2 * l + 0] = red[0][threadIdx.x][0] + red[1][threadIdx.x][0] + red[2][threadIdx.x][0] + red[3][threadIdx.x][0];
            counts[2 * l + 1] = red[0][threadIdx.x][1] + red[1][threadIdx.x][1] + red[2][threadIdx.x][1] + red[3][threadIdx.x][1];
        }
    }
}

// pair[(i * W + d) * 2 + {0,1}] = {total, HAB} of SNPs (i, i+d), d = 1 .. W-1, both inside the
// chromosome [lo, hi); d = 0 and pairs leaving the chromosome stay 0.  One workgroup per SNP i,
// threads over d: plane words of i are broadcast, those of i+d consecutive.
__global__ void __launch_bounds__(256)
ld_pair_kernel(const uint64_t *__restrict__ planeM, const uint64_t *__restrict__ planeH, int nblk,
               int64_t nloci, int64_t lo, int64_t hi, int W, int32_t *__restrict__ pair)
{
    const int64_t i = lo + blockIdx.x;
    for (int d = 1 + threadIdx.x; d < W; d += blockDim.x) {
        const int64_t j = i + d;
        int32_t tot = 0, hab = 0;
        if (j < hi) {
            for (int blk = 0; blk < nblk; blk++) {
                const int64_t base = (int64_t)blk * nloci;
                tot += __popcll(planeM[base + i] & planeM[base + j]);
                hab += __popcll(planeH[base + i] & planeH[base + j]);
            }
        }
        pair[(i * W + d) * 2 + 0] = tot;
        pair[(i * W + d) * 2 + 1] = hab;
    }
}

// The same counts from LDS: ld_pair_kernel streams four plane words per (pair, block) from L2 and is
// bound by that (30 % of the LD time at 10M SNPs x 1250).  Here one workgroup owns LD_PAIR_T
// consecutive SNPs i and thread d-1 their partners i + d: the {M, H} words of the SNPs
// i0 .. i0+T+W-2 are staged LD_PAIR_BLK blocks at a time, the counts of the thread's T pairs stay in
// registers across the chunks.  Partner words: consecutive threads, consecutive 16-B entries; the
// SNP's own words: one broadcast.
constexpr int LD_PAIR_T = 32;
constexpr int LD_PAIR_BLK = 8;
// PHASED: four planes per SNP -- {M, T ("genotype 2"), O ("genotype 1"), F (firstCopy)} -- and the
// counts {2 * #(both non-missing), x11} of r2 (garlic-data.cpp:592-606)
struct LdPairChr {       // one chromosome's share of the grid (all chromosomes in one launch)
    int64_t lo, hi;      // its SNPs
    int64_t block0;      // its first workgroup
};
template <bool PHASED>
__global__ void __launch_bounds__(256)
ld_pair_tiled_kernel(const uint64_t *__restrict__ planeM, const uint64_t *__restrict__ planeH,
                     const uint64_t *__restrict__ planeO, const uint64_t *__restrict__ planeF, int nblk,
                     int64_t nloci, const LdPairChr *__restrict__ chrs, int nchr, int W, int32_t *__restrict__ pair)
{
    constexpr int NP = PHASED ? 4 : 2;
    extern __shared__ uint64_t ld_planes[];                    // [LD_PAIR_BLK][T + W - 1][NP]
    const int span = LD_PAIR_T + W - 1;
    int c = 0;
    while (c + 1 < nchr && (int64_t)blockIdx.x >= chrs[c + 1].block0) c++;
    const int64_t lo = chrs[c].lo, hi = chrs[c].hi;
    const int64_t i0 = lo + ((int64_t)blockIdx.x - chrs[c].block0) * LD_PAIR_T;
    const int ni = (int)min<int64_t>(LD_PAIR_T, hi - i0);      // SNPs i of this tile
    const int nsnp = (int)min<int64_t>(span, hi - i0);         // staged SNPs that exist
    const int d = 1 + (int)threadIdx.x;                        // blockDim.x >= W - 1
    int32_t tot[LD_PAIR_T], hab[LD_PAIR_T];
#pragma unroll
    for (int q = 0; q < LD_PAIR_T; q++) { tot[q] = 0; hab[q] = 0; }
    for (int b0 = 0; b0 < nblk; b0 += LD_PAIR_BLK) {
        const int nb = min(LD_PAIR_BLK, nblk - b0);
        __syncthreads();
        for (int b = 0; b < nb; b++)
            for (int x = threadIdx.x; x < nsnp; x += blockDim.x) {
                const int64_t g = (int64_t)(b0 + b) * nloci + i0 + x;
                uint64_t *e = ld_planes + (size_t)(b * span + x) * NP;
                e[0] = planeM[g];
                e[1] = planeH[g];
                if (PHASED) { e[2] = planeO[g]; e[3] = planeF[g]; }
            }
        __syncthreads();
        if (d < W) {
#pragma unroll
            for (int q = 0; q < LD_PAIR_T; q++) {
                if (q + d >= nsnp) continue;                   // partner outside the chromosome: stays 0
                int32_t t = 0, h = 0;
                for (int b = 0; b < nb; b++) {
                    // (the SNP's own words through the scalar cache instead of the broadcast LDS read: measured, 11.7 ->
                    // 16.4 ms -- the scalar loads are waited for where they are issued)
                    const uint64_t *a = ld_planes + (size_t)(b * span + q) * NP, *c = a + (size_t)d * NP;
                    const ulonglong2 a0 = *reinterpret_cast<const ulonglong2 *>(a);
                    const ulonglong2 c0 = *reinterpret_cast<const ulonglong2 *>(c);
                    if (PHASED) {
                        const ulonglong2 a1 = *reinterpret_cast<const ulonglong2 *>(a + 2);
                        const ulonglong2 c1 = *reinterpret_cast<const ulonglong2 *>(c + 2);
                        t += 2 * __popcll(a0.x & c0.x);
                        h += 2 * __popcll(a0.y & c0.y) + __popcll(a1.x & c0.y) + __popcll(a0.y & c1.x) +
                             __popcll(a1.x & c1.x & ~(a1.y ^ c1.y));
                    } else {
                        t += __popcll(a0.x & c0.x);
                        h += __popcll(a0.y & c0.y);
                    }
                }
                tot[q] += t;
                hab[q] += h;
            }
        }
    }
    if (d < W) {
#pragma unroll
        for (int q = 0; q < LD_PAIR_T; q++) {
            if (q >= ni) continue;
            pair[((i0 + q) * W + d) * 2 + 0] = tot[q];           // pairs leaving the chromosome: 0
            pair[((i0 + q) * W + d) * 2 + 1] = hab[q];
            if (d == 1) {                                        // d = 0 is not a pair: 0 (the table needs no zero-fill)
                pair[((i0 + q) * W) * 2 + 0] = 0;
                pair[((i0 + q) * W) * 2 + 1] = 0;
            }
        }
    }
}

// --phased pair counts: pair = {2 * #(both non-missing), x11}  (r2, garlic-data.cpp:592-606)
__global__ void __launch_bounds__(256)
ld_pair_phased_kernel(const uint64_t *__restrict__ planeM, const uint64_t *__restrict__ planeT,
                      const uint64_t *__restrict__ planeO, const uint64_t *__restrict__ planeF, int nblk,
                      int64_t nloci, int64_t lo, int64_t hi, int W, int32_t *__restrict__ pair)
{
    const int64_t i = lo + blockIdx.x;
    for (int d = 1 + threadIdx.x; d < W; d += blockDim.x) {
        const int64_t j = i + d;
        int32_t tot = 0, x11 = 0;
        if (j < hi) {
            for (int blk = 0; blk < nblk; blk++) {
                const int64_t base = (int64_t)blk * nloci;
                const uint64_t ti = planeT[base + i], tj = planeT[base + j];
                const uint64_t oi = planeO[base + i], oj = planeO[base + j];
                tot += 2 * __popcll(planeM[base + i] & planeM[base + j]);
                x11 += 2 * __popcll(ti & tj) + __popcll(oi & tj) + __popcll(ti & oj) +
                       __popcll(oi & oj & ~(planeF[base + i] ^ planeF[base + j]));
            }
        }
        pair[(i * W + d) * 2 + 0] = tot;
        pair[(i * W + d) * 2 + 1] = x11;
    }
}

// HapData::firstCopy rows (uint8 [count][ld], loci [l0, l0 + count)) -> bit planes [blk][nloci];
// one wave per (locus, 64-individual block)
__global__ void __launch_bounds__(256)
phase_planes_kernel(const uint8_t *__restrict__ fc, int64_t ld, int64_t l0, int64_t count, int nind,
                    int nblk, int64_t nloci, uint64_t *__restrict__ planeF)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= count * nblk) return;
    const int64_t r = wave / nblk;
    const int blk = (int)(wave % nblk);
    const int ind = blk * WAVE + lane;
    const uint64_t bits = __ballot(ind < nind && fc[r * ld + ind] != 0);
    if (lane == 0) planeF[(int64_t)blk * nloci + l0 + r] = bits;
}

__global__ void ld_homfreq_kernel(const int32_t *__restrict__ counts, int64_t nloci, double *__restrict__ hf)
{
    const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nloci) return;
    double hom = (double)counts[2 * l], total = (double)counts[2 * l + 1];
    hom /= total;                                   // 0/0 = NaN as on the host: such SNPs give hr2 = 0
    hf[l] = x86_nan_if_nan(hom);
}

// garlic-data.cpp:558-583 with the two counts already taken
__device__ __forceinline__ double hr2_from_counts(double HA, double HB, int32_t hab, int32_t tot)
{
    if (!(HA > 0 && HA < 1 && HB > 0 && HB < 1)) return 0.0;
    double HAB = (double)hab;
    const double total = (double)tot;
    HAB /= total;
    const double H = HAB - HA * HB;
    const double v = H * H / (HA * (1 - HA) * HB * (1 - HB));
    return (v > 1) ? 1.0 : x86_nan_if_nan(v);     // HA, HB are finite here: a NaN is the 0/0 of HAB /= total
}

// fwd[i * W + d] = hr2(i, i + d);  bwd[(i + d) * W + d] = hr2(i + d, i)      (d = 1 .. W-1)
// COMBINED (for ld_sum_col_kernel): one row of 2W doubles per SNP in `fwd`,
//     C[i][W-1 + d] = hr2(i, i + d), d in (-W, W): the value of the pair (i, i+d) as SNP i's term; C[i][W-1] = 1.0
template <bool COMBINED>
__global__ void __launch_bounds__(256)
ld_hr2_kernel(const int32_t *__restrict__ pair, const double *__restrict__ hf, int64_t lo, int64_t hi,
              int W, double *__restrict__ fwd, double *__restrict__ bwd)
{
    const int64_t i = lo + blockIdx.x;
    const double HA = hf[i];
    if (COMBINED && threadIdx.x == 0) fwd[i * 2 * W + W - 1] = 1.0;
    for (int d = 1 + threadIdx.x; d < W; d += blockDim.x) {
        const int64_t j = i + d;
        if (j >= hi) continue;
        const int32_t tot = pair[(i * W + d) * 2], hab = pair[(i * W + d) * 2 + 1];
        const double HB = hf[j];
        if (COMBINED) {
            fwd[i * 2 * W + W - 1 + d] = hr2_from_counts(HA, HB, hab, tot);
            fwd[j * 2 * W + W - 1 - d] = hr2_from_counts(HB, HA, hab, tot);
        } else {
            fwd[i * W + d] = hr2_from_counts(HA, HB, hab, tot);
            bwd[j * W + d] = hr2_from_counts(HB, HA, hab, tot);
        }
    }
}

// Pair counts with a lane per SNP.  ld_pair_tiled_kernel gives a thread to a distance: every (SNP, block) costs it
// two LDS reads -- the SNP's own words, broadcast, and the partner's -- and it runs at the pace of those (0.35 of the
// AND + popcount rate), with nine lanes of a wave at work when W = 10.  Here thread = SNP i of a tile of 256: its own
// words are read once per block, the partner words of SNP i + d are the neighbouring lanes' entries (consecutive
// 16-B reads, conflict-free), 16 or 32 distances at a time in registers.  The plane words of the tile's 256 + W - 1
// SNPs are staged `nb_stage` blocks at a time (a few: 22 KB of LDS at W = 100, six workgroups per CU) and staged again
// for every pass over the distances (from L2: 3 passes at W = 100; keeping all blocks resident instead left one
// workgroup per CU and ran at half the pace).  SNPs past the chromosome are staged as zero words: their pairs count 0, as the table wants.
constexpr int LD_LANE_T = 256;      // distances per pass: 16 (narrow windows: one pass) or 32
template <bool PHASED, int LD_LANE_DC>
__global__ void __launch_bounds__(LD_LANE_T)
ld_pair_lane_kernel(const uint64_t *__restrict__ planeM, const uint64_t *__restrict__ planeH,
                    const uint64_t *__restrict__ planeO, const uint64_t *__restrict__ planeF, int nblk, int64_t nloci,
                    const LdPairChr *__restrict__ chrs, int nchr, int W, int nb_stage, int32_t *__restrict__ pair)
{
    constexpr int NP = PHASED ? 4 : 2;
    extern __shared__ uint64_t ld_planes[];                    // [nb_stage][span][NP]
    const int span = LD_LANE_T + W - 1;
    int c = 0;
    while (c + 1 < nchr && (int64_t)blockIdx.x >= chrs[c + 1].block0) c++;
    const int64_t hi = chrs[c].hi;
    const int64_t i0 = chrs[c].lo + ((int64_t)blockIdx.x - chrs[c].block0) * LD_LANE_T;
    const int nsnp = (int)min<int64_t>(span, hi - i0);         // staged SNPs that exist
    const int i = threadIdx.x;
    const bool mine = i0 + i < hi;
    for (int d0 = 1; d0 < W; d0 += LD_LANE_DC) {
        int32_t tot[LD_LANE_DC], hab[LD_LANE_DC];
#pragma unroll
        for (int r = 0; r < LD_LANE_DC; r++) { tot[r] = 0; hab[r] = 0; }
        for (int b0 = 0; b0 < nblk; b0 += nb_stage) {
            const int nb = min(nb_stage, nblk - b0);
            if (d0 == 1 || nb_stage < nblk) {                  // (all blocks resident: staged once)
                __syncthreads();
                for (int b = 0; b < nb; b++)
                    for (int x = threadIdx.x; x < span; x += LD_LANE_T) {
                        const int64_t g = (int64_t)(b0 + b) * nloci + i0 + x;
                        uint64_t *e = ld_planes + (size_t)(b * span + x) * NP;
                        const bool in = x < nsnp;
                        e[0] = in ? planeM[g] : 0;
                        e[1] = in ? planeH[g] : 0;
                        if (PHASED) { e[2] = in ? planeO[g] : 0; e[3] = in ? planeF[g] : 0; }
                    }
                __syncthreads();
            }
            for (int b = 0; b < nb; b++) {
                const uint64_t *a = ld_planes + (size_t)(b * span + i) * NP;
                const ulonglong2 a0 = *reinterpret_cast<const ulonglong2 *>(a);
                ulonglong2 a1 = make_ulonglong2(0, 0);
                if (PHASED) a1 = *reinterpret_cast<const ulonglong2 *>(a + 2);
#pragma unroll
                for (int r = 0; r < LD_LANE_DC; r++) {
                    if (d0 + r >= W) break;                    // wave-uniform
                    const uint64_t *cp = a + (size_t)(d0 + r) * NP;
                    const ulonglong2 c0 = *reinterpret_cast<const ulonglong2 *>(cp);
                    if (PHASED) {
                        const ulonglong2 c1 = *reinterpret_cast<const ulonglong2 *>(cp + 2);
                        tot[r] += 2 * __popcll(a0.x & c0.x);
                        hab[r] += 2 * __popcll(a0.y & c0.y) + __popcll(a1.x & c0.y) + __popcll(a0.y & c1.x) +
                                  __popcll(a1.x & c1.x & ~(a1.y ^ c1.y));
                    } else {
                        tot[r] += __popcll(a0.x & c0.x);
                        hab[r] += __popcll(a0.y & c0.y);
                    }
                }
            }
        }
        if (mine) {
            int32_t *row = pair + (i0 + i) * (int64_t)W * 2;
            if (d0 == 1) *reinterpret_cast<int2 *>(row) = make_int2(0, 0);       // d = 0 is not a pair
#pragma unroll
            for (int r = 0; r < LD_LANE_DC; r++)
                if (d0 + r < W) *reinterpret_cast<int2 *>(row + (d0 + r) * 2) = make_int2(tot[r], hab[r]);
        }
    }
}

// Narrow windows (W <= LD_SMALL_MAX_W; GARLIC's default --winsize is 10): the tiled kernels above give a thread to
// every distance or column and a workgroup to a few SNPs -- at W = 10 nine lanes of a wave work.  Flat forms:
// one thread per (SNP, distance) for the pair counts, one per (window start, column) for the sums, the plane words
// and the pair counts straight from memory (neighbouring threads share them: L1 / L2 hits).
constexpr int LD_SMALL_MAX_W = 16;     // (the in-place hr2 evaluations grow with W^2: 3.5 ms at W = 10, 31 ms at 32)

template <bool PHASED>
__global__ void __launch_bounds__(256)
ld_pair_flat_kernel(const uint64_t *__restrict__ planeM, const uint64_t *__restrict__ planeH,
                    const uint64_t *__restrict__ planeO, const uint64_t *__restrict__ planeF, int nblk, int64_t nloci,
                    const int64_t *__restrict__ chr_off, int nchr, int W, int32_t *__restrict__ pair)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // (i, d), d = 0 .. W-1
    if (e >= nloci * W) return;
    const int64_t i = e / W;
    const int d = (int)(e - i * W);
    const int64_t j = i + d;
    int32_t tot = 0, hab = 0;
    if (d > 0 && j < nloci && j < chr_off[find_chr(chr_off, nchr, i) + 1]) {   // pairs leaving the chromosome stay 0
        for (int blk = 0; blk < nblk; blk++) {
            const int64_t base = (int64_t)blk * nloci;
            if (PHASED) {
                const uint64_t ti = planeH[base + i], tj = planeH[base + j], oi = planeO[base + i], oj = planeO[base + j];
                tot += 2 * __popcll(planeM[base + i] & planeM[base + j]);
                hab += 2 * __popcll(ti & tj) + __popcll(oi & tj) + __popcll(ti & oj) +
                       __popcll(oi & oj & ~(planeF[base + i] ^ planeF[base + j]));
            } else {
                tot += __popcll(planeM[base + i] & planeM[base + j]);
                hab += __popcll(planeH[base + i] & planeH[base + j]);
            }
        }
    }
    *reinterpret_cast<int2 *>(pair + e * 2) = make_int2(tot, hab);
}

// LD[s][k] = sum over the window's SNPs i = s .. s+W-1, in that order from 0.0, of hr2(i, s + k) (1.0 for i = s + k):
// garlic-data.cpp:521-527 with hr2 / r2 evaluated in place from the pair counts (W^2 evaluations per window start: at
// W <= 32 cheaper than a table of them and the passes that write and read it).  Window starts without a full window
// keep initLDData's zeros (the caller clears the table).
__global__ void __launch_bounds__(256)
ld_sum_flat_kernel(const int32_t *__restrict__ pair, const double *__restrict__ hf, const int64_t *__restrict__ chr_off,
                   int nchr, int64_t nloci, int W, double *__restrict__ ld)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // (s, k)
    if (e >= nloci * W) return;
    const int64_t s = e / W;
    const int k = (int)(e - s * W);
    if (s + W > chr_off[find_chr(chr_off, nchr, s) + 1]) return;
    const int64_t t = s + k;
    const double HT = hf[t];
    double acc = 0.0;
    for (int64_t i = s; i < s + W; i++) {
        double term = 1.0;
        if (i < t) {
            const int2 c = *reinterpret_cast<const int2 *>(pair + (i * W + (t - i)) * 2);
            term = hr2_from_counts(hf[i], HT, c.y, c.x);
        } else if (i > t) {
            const int2 c = *reinterpret_cast<const int2 *>(pair + (t * W + (i - t)) * 2);
            term = hr2_from_counts(hf[i], HT, c.y, c.x);
        }
        acc += term;
    }
    ld[e] = x86_nan_if_nan(acc);
}

// The combined table from LDS tiles.  ld_hr2_kernel<true> writes hr2(i + d, i) into row i + d with one 8-B store
// per pair, every lane to another row (14.5 ms of the LD call at 10M SNPs x 1250, W = 100, for 24 GB of traffic).
// Here a workgroup owns LD_HR2_T consecutive SNPs i: the pair counts are read and the values of row i written
// along d as before; the values for the partner rows go through an LDS tile [i][d] and leave as pieces of
// LD_HR2_T consecutive doubles per row (C[j][W-1 - (j - i)], i ascending).  The two values of a pair share
// HAB / total and H * H -- the same operations in the same order as two calls of hr2_from_counts.
#ifndef GARLIC_LD_HR2_T
#define GARLIC_LD_HR2_T 32
#endif
constexpr int LD_HR2_T = GARLIC_LD_HR2_T;
__global__ void __launch_bounds__(256)
ld_hr2_tile_kernel(const int32_t *__restrict__ pair, const double *__restrict__ hf, int64_t lo, int64_t hi,
                   int W, double *__restrict__ C)
{
    extern __shared__ double hr2_lds[];               // hf of the SNPs i0 .. i0 + T + W - 2, then the tile [T][pitch]
    const int pitch = W + (W & 1);                    // odd distance between the lanes' reads of the second pass
    double *hfl = hr2_lds, *tile = hr2_lds + LD_HR2_T + W;
    const int64_t i0 = lo + (int64_t)blockIdx.x * LD_HR2_T;
    for (int e = threadIdx.x; e < LD_HR2_T + W - 1; e += blockDim.x) hfl[e] = (i0 + e < hi) ? hf[i0 + e] : 0.0;
    __syncthreads();
    for (int e = threadIdx.x; e < LD_HR2_T * W; e += blockDim.x) {
        const int ii = e / W, d = e - ii * W;
        const int64_t i = i0 + ii, j = i + d;
        if (j >= hi) continue;                        // (i <= j)
        if (d == 0) { C[i * 2 * W + W - 1] = 1.0; continue; }
        const int2 cnt = *reinterpret_cast<const int2 *>(pair + (i * W + d) * 2);     // {total, HAB}
        const double HA = hfl[ii], HB = hfl[ii + d];
        double f = 0.0, b = 0.0;
        if (HA > 0 && HA < 1 && HB > 0 && HB < 1) {   // hr2_from_counts(HA, HB, ..) and (HB, HA, ..)
            double HAB = (double)cnt.y;
            HAB /= (double)cnt.x;
            const double H = HAB - HA * HB, HH = H * H;   // (HB * HA is the same product)
            const double vf = HH / (HA * (1 - HA) * HB * (1 - HB));
            const double vb = HH / (HB * (1 - HB) * HA * (1 - HA));
            f = (vf > 1) ? 1.0 : x86_nan_if_nan(vf);
            b = (vb > 1) ? 1.0 : x86_nan_if_nan(vb);
        }
        C[i * 2 * W + W - 1 + d] = f;
        tile[ii * pitch + d] = b;
    }
    __syncthreads();
    // row j = i0 + 1 + jj takes tile[ii][j - i] for the SNPs i = i0 + ii of the tile with 1 <= j - i < W
    const int nrows = LD_HR2_T + W - 2;
    for (int e = threadIdx.x; e < nrows * LD_HR2_T; e += blockDim.x) {
        const int jj = e / LD_HR2_T, ii = e - jj * LD_HR2_T;
        const int d = jj + 1 - ii;
        const int64_t j = i0 + 1 + jj;
        if (d < 1 || d >= W || j >= hi) continue;
        C[j * 2 * W + W - 1 - d] = tile[ii * pitch + d];
    }
}

// LD[s][k], one workgroup per window start s (threads over k), s <= hi - W
__global__ void __launch_bounds__(256)
ld_sum_kernel(const double *__restrict__ fwd, const double *__restrict__ bwd, int64_t lo, int W,
              double *__restrict__ ld)
{
    const int64_t s = lo + blockIdx.x;
    for (int k = threadIdx.x; k < W; k += blockDim.x) {
        const int64_t t = s + k;
        double acc = 0.0;
        for (int64_t i = s; i < s + W; i++) {
            double term = 1.0;                               // i == site: += 1 (garlic-data.cpp:524)
            if (i < t) term = fwd[i * W + (t - i)];
            else if (i > t) term = bwd[i * W + (i - t)];
            acc += term;
        }
        ld[s * W + k] = x86_nan_if_nan(acc);
    }
}

// The same sums with the W^2 terms of a window start read from LDS instead of L2 (ld_sum_kernel
// moves 16 W^2 bytes per window start through L2 and is bound by that: 21.6 ms of a 39-ms call at
// 2M SNPs x 1280).  One workgroup = LD_SUM_B consecutive window starts s0 .. s0+B-1; thread k holds
// the B accumulators LD[s0+q][k] in registers.  The workgroup walks the SNPs i = s0, s0+1, ..; per
// step it stages ONE combined row
//     C_i[x] = hr2(i, i + d),  d = x - (W-1) in (-W, W):   bwd[i][-d] | 1.0 | fwd[i][d]
// (2W-1 doubles, double-buffered) and every accumulator whose window contains i -- a wave-uniform
// condition on q -- adds C_i[(s0+q+k) - i + W-1]: consecutive lanes read consecutive doubles.  The
// terms of one accumulator arrive in the order i = s .. s+W-1, from 0.0, as in ldHR2
// (garlic-data.cpp:521-527).
constexpr int LD_SUM_B = 32;   // 64 keeps the ramps shorter but costs occupancy (128 registers of accumulators): 19.4 vs 17.2 ms
constexpr int LD_SUM_AHEAD = 4;
constexpr int LD_SUM_MAX_W = 256;                              // one thread per column
struct LdSumChr {        // one chromosome's share of the grid
    int64_t lo;          // global index of its first SNP
    int64_t nstarts;     // window starts with a full window (>= 1)
    int64_t block0;      // first workgroup of the chromosome
};
__global__ void __launch_bounds__(LD_SUM_MAX_W)
ld_sum_tiled_kernel(const double *__restrict__ fwd, const double *__restrict__ bwd,
                    const LdSumChr *__restrict__ chrs, int nchr, int W, double *__restrict__ ld)
{
    // two row buffers of L = 64 + (2W-1) + 64 + blockDim doubles: the row proper sits 64 doubles in,
    // so that every accumulator of every lane -- whether its window contains the step's SNP or not,
    // lanes k >= W included -- reads inside its buffer and the look-ups need no clamping
    extern __shared__ double ld_rows[];
    const int k = threadIdx.x, nthreads = blockDim.x, RW = 2 * W - 1, L = RW + 128 + nthreads;
    int c = 0;                                                // all chromosomes in one grid
    while (c + 1 < nchr && (int64_t)blockIdx.x >= chrs[c + 1].block0) c++;
    const int64_t s0 = chrs[c].lo + ((int64_t)blockIdx.x - chrs[c].block0) * LD_SUM_B;
    const int ns = (int)min<int64_t>(LD_SUM_B, chrs[c].lo + chrs[c].nstarts - s0);
    const int nsteps = ns + W - 1;
    auto term = [&](int64_t i, int x) -> double {            // C_i[x]
        return x < W - 1 ? bwd[i * W + (W - 1 - x)] : (x == W - 1 ? 1.0 : fwd[i * W + (x - W + 1)]);
    };
    double acc[LD_SUM_B];
#pragma unroll
    for (int q = 0; q < LD_SUM_B; q++) acc[q] = 0.0;
    // rows are requested LD_SUM_AHEAD steps before they are staged (a step is far shorter than a trip
    // to memory); at most 2 elements per thread and row (RW < 2 * nthreads)
    const int x0 = min(k, RW - 1), x1 = min(k + nthreads, RW - 1);
    double pre[LD_SUM_AHEAD][2];
#pragma unroll
    for (int u = 0; u < LD_SUM_AHEAD; u++) {
        const int64_t i = s0 + min(u, nsteps - 1);
        pre[u][0] = term(i, x0);
        pre[u][1] = term(i, x1);
    }
    for (int j0 = 0; j0 < nsteps; j0 += LD_SUM_AHEAD) {
#pragma unroll
        for (int u = 0; u < LD_SUM_AHEAD; u++) {
            const int j = j0 + u;
            if (j >= nsteps) break;
            double *row = ld_rows + (j & 1) * L + 64;
            if (k < RW) row[k] = pre[u][0];
            if (k + nthreads < RW) row[k + nthreads] = pre[u][1];
            {
                const int64_t i = s0 + min(j + LD_SUM_AHEAD, nsteps - 1);
                pre[u][0] = term(i, x0);
                pre[u][1] = term(i, x1);
            }
            __syncthreads();                                  // (the other buffer was last read two steps ago)
            // accumulators q in [qlo, qhi] take a term of this row.  Per group of 16 that touches the
            // range: all 16 look-ups in flight (immediate offsets), then the adds, each behind a
            // wave-uniform branch -- a test in front of each look-up would pay one LDS round trip per
            // term, a select per add costs twice the add (and one test per quad of adds, tried, made
            // the compiler's code slower).
            const int qlo = max(0, j - W + 1), qhi = min(j, ns - 1);
            const double *cp = row + (W - 1 + k - j);         // + q
#pragma unroll
            for (int g = 0; g < LD_SUM_B; g += 16) {
                if (g > qhi || g + 15 < qlo) continue;
                double t[16];
#pragma unroll
                for (int r = 0; r < 16; r++) t[r] = cp[g + r];
#pragma unroll
                for (int r = 0; r < 16; r++)
                    if (g + r >= qlo && g + r <= qhi) {
                        acc[g + r] += t[r];
                        asm volatile("" ::: "memory");        // keeps the branch a branch
                    }
            }
        }
    }
    if (k < W) {
#pragma unroll
        for (int q = 0; q < LD_SUM_B; q++)
            if (q < ns) ld[(s0 + q) * W + k] = x86_nan_if_nan(acc[q]);
    }
}

// The ordered sums once more, transposed: thread = the window's SNP t = s + k, accumulators = the window
// starts.  ld_sum_tiled_kernel's thread owns the column k of LD_SUM_B window starts, so each of its adds
// takes a different element of the staged row (SNP t = s0 + q + k differs with q): one 8-B LDS read per add,
// and the kernel runs at the pace of the LDS (32.5 ms of an 86-ms LD call at 10M SNPs x 1250, W = 100).  With a
// thread per SNP t the term hr2(i, t) of step i is the SAME for every window start whose window contains i
// and t: one LDS read per step and thread, up to LD_COL_B adds on it.  Thread tl (t = s0 + tl) holds
// acc[q] = LD[s0 + q][tl - q] for the starts q with 0 <= tl - q < W (the other accumulators collect values that
// are never stored); every accumulator still receives its terms in the order i = s .. s+W-1 from 0.0
// (garlic-data.cpp:521-527).  Threads: W + B - 1 SNPs rounded up to whole waves -- the host picks
// B = min(32, threads - W + 1) so that few lanes idle (W = 100: 128 threads, 29 starts per workgroup).
// The rows come from ld_hr2_kernel<true>'s combined table (one contiguous row of 2W doubles per SNP).
constexpr int LD_COL_B = 32;
constexpr int LD_COL_MAX_THREADS = 576;      // W + B - 1 SNPs in whole waves: W <= 512
#ifndef GARLIC_LD_COL_BATCH
#define GARLIC_LD_COL_BATCH 4
#endif
constexpr int LD_COL_BATCH = GARLIC_LD_COL_BATCH;      // rows per wait + barrier (8: measured, see DESIGN section 3 LD)
constexpr int LD_COL_NBATCH = 4;     // batches in the LDS ring (NBATCH - 1 requested ahead)
constexpr int LD_COL_MAX_PIECES = 5; // 1-KB requests per row (8 B per thread, 576 threads at most)

// waits until at most n of the wave's vector-memory requests are outstanding (n wave-uniform, 0 .. 63)
__device__ __forceinline__ void ld_col_wait(int n)
{
    switch (n) {
#define LD_COL_WAIT_CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        LD_COL_WAIT_CASE(0)
        LD_COL_WAIT_CASE(1)
        LD_COL_WAIT_CASE(2)
        LD_COL_WAIT_CASE(3)
        LD_COL_WAIT_CASE(4)
        LD_COL_WAIT_CASE(5)
        LD_COL_WAIT_CASE(6)
        LD_COL_WAIT_CASE(7)
        LD_COL_WAIT_CASE(8)
        LD_COL_WAIT_CASE(9)
        LD_COL_WAIT_CASE(10)
        LD_COL_WAIT_CASE(11)
        LD_COL_WAIT_CASE(12)
        LD_COL_WAIT_CASE(13)
        LD_COL_WAIT_CASE(14)
        LD_COL_WAIT_CASE(15)
        LD_COL_WAIT_CASE(16)
        LD_COL_WAIT_CASE(17)
        LD_COL_WAIT_CASE(18)
        LD_COL_WAIT_CASE(19)
        LD_COL_WAIT_CASE(20)
        LD_COL_WAIT_CASE(21)
        LD_COL_WAIT_CASE(22)
        LD_COL_WAIT_CASE(23)
        LD_COL_WAIT_CASE(24)
        LD_COL_WAIT_CASE(25)
        LD_COL_WAIT_CASE(26)
        LD_COL_WAIT_CASE(27)
        LD_COL_WAIT_CASE(28)
        LD_COL_WAIT_CASE(29)
        LD_COL_WAIT_CASE(30)
        LD_COL_WAIT_CASE(31)
        LD_COL_WAIT_CASE(32)
        LD_COL_WAIT_CASE(33)
        LD_COL_WAIT_CASE(34)
        LD_COL_WAIT_CASE(35)
        LD_COL_WAIT_CASE(36)
        LD_COL_WAIT_CASE(37)
        LD_COL_WAIT_CASE(38)
        LD_COL_WAIT_CASE(39)
        LD_COL_WAIT_CASE(40)
        LD_COL_WAIT_CASE(41)
        LD_COL_WAIT_CASE(42)
        LD_COL_WAIT_CASE(43)
        LD_COL_WAIT_CASE(44)
        LD_COL_WAIT_CASE(45)
        LD_COL_WAIT_CASE(46)
        LD_COL_WAIT_CASE(47)
        LD_COL_WAIT_CASE(48)
        LD_COL_WAIT_CASE(49)
        LD_COL_WAIT_CASE(50)
        LD_COL_WAIT_CASE(51)
        LD_COL_WAIT_CASE(52)
        LD_COL_WAIT_CASE(53)
        LD_COL_WAIT_CASE(54)
        LD_COL_WAIT_CASE(55)
        LD_COL_WAIT_CASE(56)
        LD_COL_WAIT_CASE(57)
        LD_COL_WAIT_CASE(58)
        LD_COL_WAIT_CASE(59)
        LD_COL_WAIT_CASE(60)
        LD_COL_WAIT_CASE(61)
        LD_COL_WAIT_CASE(62)
        LD_COL_WAIT_CASE(63)
#undef LD_COL_WAIT_CASE
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// acc[q] += h for q = first .. 31 (ASCENDING) or q = 31 - first .. 0 (descending), `first` wave-uniform: the 32
// adds are laid out in that order and the wave jumps over the first `first` of them (8 bytes each) -- one computed
// branch per step where a test per accumulator costs two scalar instructions each and, written in C++, made
// hipcc copy every accumulator every step.
#define LD_COL_ADD(q) "v_add_f64 %[a" #q "], %[a" #q "], %[h]\n\t"
#define LD_COL_OPS(a)                                                                                              \
    [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]), [a3] "+v"(a[3]), [a4] "+v"(a[4]), [a5] "+v"(a[5]),          \
    [a6] "+v"(a[6]), [a7] "+v"(a[7]), [a8] "+v"(a[8]), [a9] "+v"(a[9]), [a10] "+v"(a[10]), [a11] "+v"(a[11]),      \
    [a12] "+v"(a[12]), [a13] "+v"(a[13]), [a14] "+v"(a[14]), [a15] "+v"(a[15]), [a16] "+v"(a[16]),                 \
    [a17] "+v"(a[17]), [a18] "+v"(a[18]), [a19] "+v"(a[19]), [a20] "+v"(a[20]), [a21] "+v"(a[21]),                 \
    [a22] "+v"(a[22]), [a23] "+v"(a[23]), [a24] "+v"(a[24]), [a25] "+v"(a[25]), [a26] "+v"(a[26]),                 \
    [a27] "+v"(a[27]), [a28] "+v"(a[28]), [a29] "+v"(a[29]), [a30] "+v"(a[30]), [a31] "+v"(a[31])
// s_getpc_b64 yields the address of the instruction behind it; three 4-byte instructions follow before the adds
#define LD_COL_JUMP "s_getpc_b64 s[98:99]\n\ts_add_u32 s98, s98, %[off]\n\ts_addc_u32 s99, s99, 0\n\ts_setpc_b64 s[98:99]\n\t"
// one block for both orders (two blocks on the same accumulators in an if / else made hipcc copy all of them at the
// join): the descending sequence, a branch over the ascending one, the ascending sequence
__device__ __forceinline__ void ld_col_adds(double (&a)[LD_COL_B], double h, bool ascending, int first)
{
    static_assert(LD_COL_B == 32, "32 adds are written out");
    const uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane(first * 8 + (ascending ? 12 + 32 * 8 + 4 : 12));
    asm volatile(LD_COL_JUMP
                 LD_COL_ADD(31) LD_COL_ADD(30) LD_COL_ADD(29) LD_COL_ADD(28) LD_COL_ADD(27) LD_COL_ADD(26)
                 LD_COL_ADD(25) LD_COL_ADD(24) LD_COL_ADD(23) LD_COL_ADD(22) LD_COL_ADD(21) LD_COL_ADD(20)
                 LD_COL_ADD(19) LD_COL_ADD(18) LD_COL_ADD(17) LD_COL_ADD(16) LD_COL_ADD(15) LD_COL_ADD(14)
                 LD_COL_ADD(13) LD_COL_ADD(12) LD_COL_ADD(11) LD_COL_ADD(10) LD_COL_ADD(9) LD_COL_ADD(8) LD_COL_ADD(7)
                 LD_COL_ADD(6) LD_COL_ADD(5) LD_COL_ADD(4) LD_COL_ADD(3) LD_COL_ADD(2) LD_COL_ADD(1) LD_COL_ADD(0)
                 "s_branch LD_COL_END_%=\n\t"
                 LD_COL_ADD(0) LD_COL_ADD(1) LD_COL_ADD(2) LD_COL_ADD(3) LD_COL_ADD(4) LD_COL_ADD(5)
                 LD_COL_ADD(6) LD_COL_ADD(7) LD_COL_ADD(8) LD_COL_ADD(9) LD_COL_ADD(10) LD_COL_ADD(11) LD_COL_ADD(12)
                 LD_COL_ADD(13) LD_COL_ADD(14) LD_COL_ADD(15) LD_COL_ADD(16) LD_COL_ADD(17) LD_COL_ADD(18)
                 LD_COL_ADD(19) LD_COL_ADD(20) LD_COL_ADD(21) LD_COL_ADD(22) LD_COL_ADD(23) LD_COL_ADD(24)
                 LD_COL_ADD(25) LD_COL_ADD(26) LD_COL_ADD(27) LD_COL_ADD(28) LD_COL_ADD(29) LD_COL_ADD(30)
                 LD_COL_ADD(31)
                 "LD_COL_END_%=:\n\t"
                 : LD_COL_OPS(a) : [h] "v"(h), [off] "s"(off) : "s98", "s99", "scc");
}
// PIECES: 1-KB requests per row = blockDim / 128 rounded up.  Rows ahead and the count of requests the wave may
// leave outstanding while it waits for a row are compile-time (s_waitcnt takes an immediate; the per-step scalar
// work -- ring offsets advanced, not recomputed, no switch -- is on the critical path of a step: ~100 scalar
// instructions per step cost as much as the step's 32 adds).
template <int PIECES>
__global__ void __launch_bounds__(LD_COL_MAX_THREADS)
ld_sum_col_kernel(const double *__restrict__ C, const LdSumChr *__restrict__ chrs, int nchr, int W, int B,
                  double *__restrict__ ld, double *__restrict__ D, unsigned nwork)
{
    // Neighbouring workgroups read the same rows of C (a row serves (B + W - 1) / B = 4.4 of them at W = 100).  Workgroups
    // are dealt round-robin over the 8 XCDs, each with an L2 of its own: numbered plainly, the neighbours sat on eight
    // different L2s and every one of them fetched its rows from memory -- 44 GB instead of 16 at 10M SNPs, and the
    // kernel ran at that rate (its skeleton, no adds and no stores: 7.2 of 10.7 ms).  Give every XCD a contiguous
    // range of the work instead (gridDim.x is a multiple of 8).
    const unsigned per_xcd = gridDim.x >> 3;
    const unsigned vblock = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (vblock >= nwork) return;
    // Step j (SNP i = s0 + j) needs of row i only the elements x = W-1 + tl - j of the workgroup's threads: blockDim
    // consecutive doubles from a_j = (W-1 - j) rounded down to even (16-B aligned for the DMA) -- 1 KB for 128
    // threads instead of the row's 2W doubles.  Thread tl finds its element at tl + ((W-1-j) & 1).  Elements in
    // front of or behind the row proper (x < 0, x > 2W-2) are the neighbouring rows' and feed accumulators that are
    // never stored.  LDS: a ring of AHEAD + 2 rows of blockDim doubles + slack (the last thread's odd step, the last
    // request's tail).
    // Rows travel in batches of LD_COL_BATCH: one wait + one barrier per batch instead of per row (a step's 32 adds
    // take 128 cycles; a wait, a barrier and the bookkeeping around them cost more than that per step).  The ring
    // holds LD_COL_NBATCH batches; the batch requested at the start of batch k lands in the part batch k-1 just left.
    constexpr int BATCH = LD_COL_BATCH, NB = LD_COL_NBATCH, NRING = BATCH * NB, STEADY = (NB - 2) * BATCH * PIECES;
    static_assert(STEADY <= 63, "requests that can be counted");
    extern __shared__ double ld_rows[];
    const int tl = threadIdx.x, P = 2 * W;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int c = 0;
    while (c + 1 < nchr && (int64_t)vblock >= chrs[c + 1].block0) c++;
    const int64_t s0 = chrs[c].lo + ((int64_t)vblock - chrs[c].block0) * B;
    const int ns = (int)min<int64_t>(B, chrs[c].lo + chrs[c].nstarts - s0);
    const int nsteps = ns + W - 1, nbatches = (nsteps + BATCH - 1) / BATCH;
    // wave 0 streams the pieces into the ring by LDS-DMA, NB - 1 batches ahead, no registers in between
    const uint32_t ring = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double *)ld_rows;
    // (row pitch = the whole requests: a request never runs into the next row's slot, which may be the batch being read)
    constexpr uint32_t row_bytes = (uint32_t)PIECES * 1024u, ring_bytes = row_bytes * NRING;
    const uint32_t lane16 = (uint32_t)(tl & 63) * 16u;
    // state of the next request: its row in C, W-1 - (its step), its ring offset, rows left to request
    const char *req_row = reinterpret_cast<const char *>(C + s0 * P);
    int req_e = W - 1, req_left = nsteps;
    uint32_t req_off = 0;
    auto request_batch = [&]() {             // BATCH rows (fewer at the strip's end; the ring offset moves on all the same)
#pragma unroll
        for (int u = 0; u < BATCH; u++) {
            if (req_left > 0) {
                const char *g = req_row + (int64_t)(req_e & ~1) * 8;
#pragma unroll
                for (int q = 0; q < PIECES; q++)
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                                 :: "s"(ring + req_off + (uint32_t)q * 1024u), "v"(lane16), "s"(g + q * 1024) : "memory");
                req_row += (int64_t)P * 8;
                req_e--;
                req_left--;
            }
            req_off += row_bytes;
        }
        if (req_off == ring_bytes) req_off = 0;
    };
    if (wave == 0)
        for (int k = 0; k < NB - 1; k++) request_batch();
    double acc[LD_COL_B];
#pragma unroll
    for (int q = 0; q < LD_COL_B; q++) acc[q] = 0.0;
    uint32_t rd_addr = ring + (uint32_t)tl * 8u;                 // this thread's element of the batch's first row, even step
    const uint32_t rd_end = rd_addr + ring_bytes;
    uint32_t odd = (uint32_t)(W - 1) & 1u;                        // (W-1-j) & 1 of the batch's first row (BATCH is even)
    static_assert(BATCH % 2 == 0, "the parity of W-1-j repeats per batch");
    for (int k = 0; k < nbatches; k++) {
        if (wave == 0) {
            // requests retire in order: all but those of the batches requested after batch k have landed
            // (exactly: a larger count would let the wait pass before batch k is complete)
            const int rows_later = min(nsteps, (k + NB - 1) * BATCH) - (k + 1) * BATCH;   // rows of the batches k+1 .. k+NB-2
            if (rows_later == (NB - 2) * BATCH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STEADY) : "memory");
            else ld_col_wait(max(rows_later, 0) * PIECES);
        }
        __syncthreads();       // batch k is in the ring; everybody is done with batch k - 1 (its part is written next)
        if (wave == 0) request_batch();
        const int j0 = k * BATCH;
        double h[BATCH];       // hr2(s0 + j, s0 + tl), j = j0 ..
#pragma unroll
        for (int u = 0; u < BATCH; u++)
            asm volatile("ds_read_b64 %0, %1" : "=v"(h[u]) : "v"(rd_addr + (uint32_t)u * row_bytes + ((odd ^ (uint32_t)(u & 1)) * 8u)) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        rd_addr += BATCH * row_bytes;
        if (rd_addr == rd_end) rd_addr -= ring_bytes;
#pragma unroll
        for (int u = 0; u < BATCH; u++) {
            const int j = j0 + u;
#ifdef GARLIC_LDS_ABL_NO_ADDS
            if (j < nsteps && h[u] == 1.2345e-300) {
#else
            if (j < nsteps) {
#endif
                const bool leaving = j >= W;
                ld_col_adds(acc, h[u], leaving,                                 // entering: q = min(j, 31) .. 0
                            leaving ? j - W + 1 : (j < LD_COL_B - 1 ? LD_COL_B - 1 - j : 0));   // leaving: q = j-W+1 .. 31
            }
        }
    }
#ifndef GARLIC_LDS_ABL_NO_LD          // timing experiments (results wrong)
    if (ld) {                         // (NULL: the caller wants the wLOD weights only -- 8 GB less to write at 10M SNPs, W = 100)
#pragma unroll
        for (int q = 0; q < LD_COL_B; q++) {
            const int k = tl - q;
            if (q < ns && k >= 0 && k < W) ld[(s0 + q) * W + k] = x86_nan_if_nan(acc[q]);
        }
    }
#else
    if (acc[0] == 1.2345e-300) ld[s0 * W + tl] = acc[1] + acc[2] + acc[3] + acc[31] + acc[16];
#endif
#ifdef GARLIC_LDS_ABL_NO_D
    if (acc[5] != 1.2345e-300) return;
#endif
    // ... and the weight the tuned wLOD kernels read, D[l][k] = 1 / LD[l - k][k] (skew_reciprocal_kernel): SNP
    // l = s0 + tl is this thread's, its row takes the thread's values back to front.  Through an LDS tile
    // [thread][16 starts], so that 16 lanes write 128 contiguous bytes of one row (straight from the registers every
    // lane of a store went to another row: 4 ms of this kernel's 15 at 10M SNPs x 1250).
    if (!D) return;
    __syncthreads();                                          // the row ring is not read any more
    double *tile = ld_rows;
    const int nthreads = blockDim.x;
    for (int q0 = 0; q0 < LD_COL_B; q0 += 16) {
#pragma unroll
        for (int u = 0; u < 16; u++)
            if (q0 == 0) tile[tl * 17 + u] = reciprocal_x86(x86_nan_if_nan(acc[u]));
            else tile[tl * 17 + u] = reciprocal_x86(x86_nan_if_nan(acc[16 + u]));
        __syncthreads();
        for (int e = tl; e < nthreads * 16; e += nthreads) {
            const int t2 = e >> 4, u = e & 15, q = q0 + u, k = t2 - q;
            if (q < ns && k >= 0 && k < W) D[(s0 + t2) * W + k] = tile[t2 * 17 + u];
        }
        __syncthreads();
    }
}

} // namespace garlic

namespace garlic {

// Pair counts on the matrix cores.  tot(i, j) = |M_i & M_j| and HAB(i, j) = |H_i & H_j| are dot products of 0/1
// vectors over the individuals -- a banded Gram matrix M M^T, H H^T (|i - j| < W): a contraction, so it belongs on
// MFMA.  ld_pair_lane_kernel does AND + popcount per (pair, block) from LDS: 9.6 ms at 10M SNPs x 1250, W = 100;
// on v_mfma_i32_32x32x32_i8 (bits as bytes) 5.6 ms; on fp4 operands (below) 4.8 ms.
//   workgroup = 128 SNPs i (4 waves x a tile of 32) against their 128 + 32 (NJ - 1) partners j, NJ = 1 + (30 + W) / 32
//   per 64-individual block: every thread turns one SNP's plane words (1 bit per individual) into 4-bit operands
//   and stores them in MFMA fragment order [plane][tile][lane][16 B] (conflict-free both ways); each wave then reads
//   its NJ fragments per plane and issues NJ MFMAs on them (its own tile is fragment 0).  Two LDS buffers: the
//   operands of block b + 1 are made while block b is multiplied.  One tile per wave: 2 NJ accumulator tiles = 160
//   registers at W = 100 (two tiles per wave, 320, did not fit the accumulation registers and hipcc moved them in
//   and out around every MFMA: 590 v_accvgpr moves per block).  The k order inside a fragment does not matter (both
//   operands use the same one: the sum runs over all of it).  Ablated at 10M x 1250 (i8 form, tools/exp/ld_mfma_abl.sh):
//   5.5 ms; without the table's stores 4.55; without the expansion 3.4; without both 2.6.
// SNPs past the chromosome are staged as zero words (their pairs count 0, as the table wants); d = 0 is written as 0.
constexpr int LDM_TI = 128;
typedef int ldm_i32x4 __attribute__((ext_vector_type(4)));

// 4-bit operands: v_mfma_scale_f32_32x32x64_f8f6f4 on fp4 (e2m1) takes a whole 64-individual block per
// instruction in the cycles the i8 form needs for 32 individuals.  A set bit becomes the nibble 0001 = 0.5, the block
// scales are 2^0, the products 0.25 and the f32 sums exact (counts below 2^22): count = 4 * sum.  Half the MFMAs,
// half the LDS bytes, 7 instead of 10 vector instructions per 8 individuals for the expansion.
typedef int ldm_i32x8 __attribute__((ext_vector_type(8)));
typedef float ldm_f32x16 __attribute__((ext_vector_type(16)));

// 8 bits -> 8 nibbles (bit m -> nibble m = 0 / 1)
__device__ __forceinline__ uint32_t ldm_spread8(uint32_t x)
{
    uint32_t t = (x | (x << 12)) & 0x000F000Fu;
    t = (t | (t << 6)) & 0x03030303u;
    return (t | (t << 3)) & 0x11111111u;
}

// HR2 (round 4): the kernel turns its counts into the two hr2 values of every pair itself and writes them straight
// into the combined table ld_sum_col_kernel reads (C[i][W-1 + d] = hr2(i, i+d), C[j][W-1 - d] = hr2(j, i), C[i][W-1] = 1)
// -- ld_hr2_tile_kernel's arithmetic to the letter (hr2_from_counts both ways, garlic-data.cpp:558-583) -- so that the
// 8-GB pair table is neither written nor read (that kernel moved 24 GB for 1.6 ms of arithmetic: 7.1 ms of the LD call at
// 10M SNPs x 1250, W = 100).  Row i's values leave along d as they sit in the accumulator tile (32 lanes = 256
// contiguous bytes); the partner rows' values are turned through an LDS tile [j][i] per wave (the operand buffers,
// idle by then) and leave as 32 consecutive doubles of row j.  Only where the counts need no sum over shards:
// garlic_panel_compute_ld; garlic_ld_counts / garlic_ld_finish keep the table.
constexpr int LDM_XT = 33;                         // doubles per row of a wave's transposition tile (conflict-free)
constexpr size_t LDM_XT_BYTES = ((size_t)4 * 32 * LDM_XT + 256) * 8;      // + homFreq of the workgroup's (up to) 256 staged SNPs

template <int NJ, bool HR2>
__global__ void __launch_bounds__(256, 2)
ld_pair_mfma_kernel(const uint64_t *__restrict__ planeM, const uint64_t *__restrict__ planeH, int nblk, int64_t nloci,
                    const LdPairChr *__restrict__ chrs, int nchr, int W, int32_t *__restrict__ pair,
                    const double *__restrict__ hf, double *__restrict__ C)
{
    constexpr int NT = 4 + NJ - 1;                 // staged tiles of 32 SNPs (<= 8: one SNP per thread)
    static_assert(NT * 32 <= 256, "one staged SNP per thread");
    constexpr int BUF = 2 * NT * WAVE * 16;        // bytes per buffer: [plane][tile][lane][16]: 32 individuals per lane
    extern __shared__ __attribute__((aligned(16))) unsigned char ldm_lds[];      // two buffers
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int c = 0;
    while (c + 1 < nchr && (int64_t)blockIdx.x >= chrs[c + 1].block0) c++;
    const int64_t hi = chrs[c].hi;
    const int64_t i0 = chrs[c].lo + ((int64_t)blockIdx.x - chrs[c].block0) * LDM_TI;
    const bool mine = tid < NT * 32;                            // this thread stages SNP i0 + tid
    const bool in = mine && i0 + tid < hi;
    uint64_t wm = 0, wh = 0;
    auto fetch = [&](int b) {
        const int64_t g = (int64_t)b * nloci + i0 + (in ? tid : 0);
        wm = in ? planeM[g] : 0;
        wh = in ? planeH[g] : 0;
    };
    auto stage = [&](int buf) {
        if (!mine) return;
        const int tile = tid >> 5, r = tid & 31;
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {
            const uint64_t w = pl ? wh : wm;
#pragma unroll
            for (int h = 0; h < 2; h++) {                      // lane half = individuals 32 h .. 32 h + 31
                const uint32_t x = (uint32_t)(w >> (32 * h));
                unsigned char *dst = ldm_lds + buf * BUF + (((pl * NT + tile) * WAVE) + h * 32 + r) * 16;
                *reinterpret_cast<uint4 *>(dst) = make_uint4(ldm_spread8(x & 0xFFu), ldm_spread8((x >> 8) & 0xFFu),
                                                             ldm_spread8((x >> 16) & 0xFFu), ldm_spread8(x >> 24));
            }
        }
    };
    ldm_f32x16 acc[NJ][2];
#pragma unroll
    for (int q = 0; q < NJ; q++)
#pragma unroll
        for (int pl = 0; pl < 2; pl++)
#pragma unroll
            for (int k = 0; k < 16; k++) acc[q][pl][k] = 0.0f;
    fetch(0);
    stage(0);
    if (nblk > 1) fetch(1);
    __syncthreads();
    const int one = 0x7F7F7F7F;                                 // E8M0 block scales: 2^0
    for (int b = 0; b < nblk; b++) {
        const int buf = b & 1;
#ifndef GARLIC_LDM_ABL_NO_STAGE       // timing experiment (results wrong)
        if (b + 1 < nblk) stage(buf ^ 1);                       // (its words were requested a block ago)
#endif
        if (b + 2 < nblk) fetch(b + 2);
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {
            ldm_i32x8 fb[NJ];
#pragma unroll
            for (int q = 0; q < NJ; q++) {
                const ldm_i32x4 f = *reinterpret_cast<const ldm_i32x4 *>(ldm_lds + buf * BUF + (((pl * NT + wave + q) * WAVE) + lane) * 16);
                fb[q] = ldm_i32x8{f[0], f[1], f[2], f[3], 0, 0, 0, 0};
            }
#pragma unroll
            for (int q = 0; q < NJ; q++)
                acc[q][pl] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[0], fb[q], acc[q][pl], 4, 4, 0, one, 0, one);
        }
        __syncthreads();
    }
    // C layout (dtype-independent): column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const int col = lane & 31, rh = 4 * (lane >> 5);
    const int64_t it = i0 + wave * 32;
    if (HR2) {
        // homFreq of the workgroup's NT x 32 SNPs behind the waves' transposition tiles
        double *xt = reinterpret_cast<double *>(ldm_lds) + (size_t)wave * 32 * LDM_XT;     // [j][i] of one 32 x 32 tile
        double *hfs = reinterpret_cast<double *>(ldm_lds) + (size_t)4 * 32 * LDM_XT;
        if (mine) hfs[tid] = in ? hf[i0 + tid] : 0.0;
        __syncthreads();
        const int P = 2 * W;
#pragma unroll
        for (int q = 0; q < NJ; q++) {
            const int64_t j = it + q * 32 + col;
            const double HB = hfs[wave * 32 + q * 32 + col];
            const bool okB = HB > 0 && HB < 1;
            const double pB = HB * (1 - HB);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int base = (k & 3) + 8 * (k >> 2);           // il = base (lanes 0-31) or base + 4
                // no lane of this (q, k) has 0 <= d < W (wave-uniform: d = 32 q + col - il): nothing to compute or keep
                if (32 * q + 31 - base < 0 || 32 * q - (base + 4) >= W) continue;
                const int il = base + rh;
                const int64_t i = it + il;
                const int64_t d = j - i;
                const double HA = hfs[wave * 32 + il];
                double f = 0.0, b = 0.0;
                if (HA > 0 && HA < 1 && okB) {                     // hr2_from_counts(HA, HB, ..) and (HB, HA, ..)
                    double HAB = (double)(int)(acc[q][1][k] * 4.0f);
                    HAB /= (double)(int)(acc[q][0][k] * 4.0f);
                    const double H = HAB - HA * HB, HH = H * H;
                    const double vf = HH / (HA * (1 - HA) * HB * (1 - HB));
                    const double vb = HH / (pB * HA * (1 - HA));
                    f = (vf > 1) ? 1.0 : x86_nan_if_nan(vf);
                    b = (vb > 1) ? 1.0 : x86_nan_if_nan(vb);
                }
                if (i < hi && j < hi && d >= 0 && d < W) C[i * P + W - 1 + d] = d == 0 ? 1.0 : f;
                xt[col * LDM_XT + il] = b;
                __builtin_amdgcn_sched_barrier(0);                 // one pair's divisions at a time: 160 accumulator registers are live
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // (the tile is this wave's own: its LDS operations execute in order)
#pragma unroll 4
            for (int r = 0; r < 16; r++) {
                const int jl = 2 * r + (lane >> 5);
                const int64_t jj = it + q * 32 + jl, ii = it + col;
                const int64_t d = jj - ii;
                const double v = xt[jl * LDM_XT + col];
                if (ii < hi && jj < hi && d >= 1 && d < W) C[jj * P + W - 1 - d] = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < NJ; q++) {
        const int64_t j = it + q * 32 + col;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int64_t i = it + (k & 3) + 8 * (k >> 2) + rh;
            const int64_t d = j - i;
            if (i < hi && d >= 0 && d < W) {
                int2 v = make_int2((int)(acc[q][0][k] * 4.0f), (int)(acc[q][1][k] * 4.0f));
                if (d == 0) v = make_int2(0, 0);
#ifdef GARLIC_LDM_ABL_NO_STORE      // timing experiment (results wrong)
                if (v.x != 0x7FFFFFF1) continue;
#endif
                *reinterpret_cast<int2 *>(pair + (i * W + d) * 2) = v;
            }
        }
    }
}

} // namespace garlic
