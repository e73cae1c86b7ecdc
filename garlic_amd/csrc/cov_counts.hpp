// The sliding coverage counts from one-bit-per-window rows (the inWin[] loop of assembleROHWindows,
// src/garlic-roh.cpp:446-454): inWin[l] = the number of set bits in (l - W, l].  Shared by cov_counts_from_bits_kernel
// (coverage_kernel.hpp: a launch of its own) and by the count items of lod_bits_kernel (feed_kernel.hpp: the same
// work taken from the chain kernel's queue while its longest runs are still under way).
#pragma once
#include "lod_kernels.hpp"

namespace garlic {

// a count item of lod_bits_kernel: COV_ITEM_WORDS 32-SNP words of COV_ITEM_ROWS individuals of one chromosome
constexpr int COV_ITEM_WORDS = 256, COV_ITEM_ROWS = 8;
// a wave's transpose buffer: 4 KB of counts as 256 uint4, one pad slot per sixteen (the writes go to slots 4 lane + u, the
// reads to u 64 + lane: without the pad sixteen lanes of a write hit four banks)
constexpr int COV_XPOSE_SLOTS = 4 * WAVE + 4 * WAVE / 16;
__device__ __forceinline__ int cov_xpose_slot(int i) { return i + (i >> 4); }

// the chromosome of word column g (word_base: prefix sums of the chromosomes' word counts).  The workgroup's first
// column is looked up once, wave-uniformly (scalar instructions); a thread then moves on by the chromosomes its own
// column lies beyond -- none, almost always: a search per thread was a dozen dependent loads in front of every thread
__device__ __forceinline__ int cov_word_chr(const int32_t *__restrict__ word_base, int nchr, int g)
{
    const int g0 = (int)(blockIdx.x * blockDim.x);
    int lo = 0, hi = nchr - 1;                 // the last chromosome with word_base[c] <= g0
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (word_base[mid] <= g0) lo = mid;
        else hi = mid - 1;
    }
    int chr = __builtin_amdgcn_readfirstlane(lo);
    while (chr + 1 < nchr && g >= word_base[chr + 1]) chr++;
    return chr;
}

// One thread, one (individual, 32-SNP word): the count of the W bits in front of the word (popcounts over W / 32 + 1
// words), then bit in, bit out, 32 times.  The 32 counts of a thread are 64 contiguous bytes of the individual's row;
// a wave whose 64 words are whole words of one chromosome (`wave_whole`: decided by the caller, the same in every
// lane) passes them through 4 KB of LDS of its own (`xw`, COV_XPOSE_SLOTS uint4) so that every store instruction writes 1 KB contiguously
// (four 16-byte pieces 64 B apart per lane otherwise).
//   brow / nwords: the individual's bit row of the chromosome; t: the thread's word (live: it exists);
//   orow: the individual's count row, nloci counts; whole: 32 t + 32 <= nloci and the layout allows 16-byte stores
__device__ __forceinline__ void cov_counts_word(const uint32_t *__restrict__ brow, int nwords, int t, bool live, int W,
                                                int16_t *__restrict__ orow, int nloci, bool whole, bool wave_whole,
                                                uint4 *xw, int lane)
{
    auto word = [&](int x) -> uint32_t { return (live && x >= 0 && x < nwords) ? brow[x] : 0u; };
    const uint32_t cur = word(t);
    const int rel = 32 * t - W, dA = rel >> 5, r = rel & 31;          // bit 32 t - W sits in word dA at bit r (floor)
    const uint32_t wA = word(dA);
    const uint32_t F = __builtin_amdgcn_alignbit(word(dA + 1), wA, (uint32_t)r);      // bit j = window 32 t - W + j
    int cnt = __popc(wA >> r);                                        // windows 32 t - W .. 32 t - 1
    for (int x = dA + 1; x < t; x += 4) {                             // (four loads in flight, not one round trip per word)
        const uint32_t w0 = word(x), w1 = x + 1 < t ? word(x + 1) : 0u, w2 = x + 2 < t ? word(x + 2) : 0u,
                       w3 = x + 3 < t ? word(x + 3) : 0u;
        cnt += (int)(__popc(w0) + __popc(w1) + __popc(w2) + __popc(w3));
    }
    uint32_t pk[16];
#pragma unroll
    for (int j = 0; j < 32; j++) {
        cnt += (int)((cur >> j) & 1u) - (int)((F >> j) & 1u);
        const uint32_t v = (uint32_t)cnt & 0xFFFFu;
        if (j & 1) pk[j >> 1] |= v << 16;
        else pk[j >> 1] = v;
    }
    int16_t *o16 = orow + 32 * t;
    if (wave_whole) {
#pragma unroll
        for (int u = 0; u < 4; u++) xw[cov_xpose_slot(4 * lane + u)] = make_uint4(pk[4 * u], pk[4 * u + 1], pk[4 * u + 2], pk[4 * u + 3]);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint4 *o = reinterpret_cast<uint4 *>(o16 - 32 * lane);       // the wave's first word
#pragma unroll
        for (int u = 0; u < 4; u++) {      // (streamed: nothing reads the counts on the device)
            const uint4 v = xw[cov_xpose_slot(u * WAVE + lane)];
            uint32_t *q = reinterpret_cast<uint32_t *>(o + u * WAVE + lane);
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4 *>(q));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // (xw is written again by the caller's next row)
        __builtin_amdgcn_wave_barrier();
        return;
    }
    if (!live) return;
    if (whole) {
        uint4 *o = reinterpret_cast<uint4 *>(o16);
#pragma unroll
        for (int u = 0; u < 4; u++) o[u] = make_uint4(pk[4 * u], pk[4 * u + 1], pk[4 * u + 2], pk[4 * u + 3]);
    } else {
#pragma unroll
        for (int j = 0; j < 32; j++)
            if (32 * t + j < nloci) o16[j] = (int16_t)((pk[j >> 1] >> (16 * (j & 1))) & 0xFFFFu);
    }
}

} // namespace garlic
