// TGLS (per-genotype likelihood) and wLOD (gap-weighted) variants of the window LOD path.
// Round-1 versions: compiler-scheduled, one wavefront per (run, 64 individuals) item, written
// for exactness first (every FP64 operation in the reference's order); the unweighted --error path
// in lod_kernels.hpp is the tuned one.
//
//   TGLS  src/garlic-roh.cpp:68,91-95,117 : error := GL[locus][ind] inside lod().  lod() needs the
//         host libm's log10, so the distinct error values of a panel are dictionary-encoded
//         (one byte per genotype) and the host tabulates lod(g, freq[l], error[code]) per SNP:
//         term = tabgl[(l * ncodes + code) * 4 + g].
//   wLOD  src/garlic-roh.cpp:204-277 : score[l] = (lod * nomut(dP[l])) * norec(dG[l]);
//         win[s] = sum_{j<W} score[s+j] * (1.0 / LD[s][j]), left to right from +0.0, every valid
//         window independently.  nomut/norec (host libm exp) come as one per-SNP pair; 1.0/LD is an
//         IEEE division done once on the device.
#pragma once
#include "lod_kernels.hpp"
#include "tgls_math.hpp"
#include "wlod_loop_gfx950.inc"


namespace garlic {

struct VariantArgs {
    const uint32_t *packed;
    const double *tab;        // [GOFF+nloci+pad][4]            (!use_gl)
    const double *tabgl;      // [GOFF+nloci+pad][ncodes][4]    (use_gl)
    const uint8_t *codes;     // [GOFF+nloci+pad][nind_pad]     (use_gl)
    const double *decay;      // [GOFF+nloci+pad][2] = {nomut, norec} (wLOD)
    const double *rld;        // [nloci][winsize] = 1.0 / LD    (wLOD)
    const ChainItem *items;
    const ChrDev *chrs;
    double *out;
    int64_t nind_pad;
    int64_t nwordrows;
    int32_t ind_begin, ind_count, winsize, ncodes, use_gl;
    const double *terms;      // use_gl with continuous likelihoods: raw term matrix [blk][term_rows][64], else NULL
    int64_t term_rows;
};

// per-SNP term of this lane's individual; G = padded global locus index
__device__ __forceinline__ double variant_term(const VariantArgs &p, int64_t G, int64_t col)
{
    if (p.use_gl && p.terms) return p.terms[((col >> 6) * p.term_rows + G) * WAVE + (col & 63)];
    const uint32_t word = p.packed[packed_index(G >> 4, col, p.nwordrows)];
    const uint32_t g = (word >> (2 * (int)(G & 15))) & 3u;
    if (p.use_gl) {
        const uint32_t code = p.codes[G * p.nind_pad + col];
        return p.tabgl[((G * p.ncodes) + code) * 4 + g];
    }
    return p.tab[G * 4 + g];
}

// masked transposed store of one 32-step tile held in LDS (64 rows x TPITCH doubles): 4 rows x 256
// contiguous bytes per instruction; one non-temporal 16-B store per lane where the layout allows
// (8-B-per-lane stores block the issuing wave ~100 cycles each -- and one wave works through a
// run's tiles one after the other, so that is on the kernel's critical path)
typedef double f64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void variant_store(const double *tile, int s0, int a, int b, int lane,
                                              int rows_valid, double *out_tile, int64_t pitch)
{
    const int rsub = lane >> 4, csub = lane & 15;
    const int s = s0 + 2 * csub;
    const bool in0 = (s >= a && s <= b), in1 = (s + 1 >= a && s + 1 <= b);
    const bool al16 = ((reinterpret_cast<uintptr_t>(out_tile) & 15) == 0) && ((pitch & 1) == 0);
    for (int q = 0; q < WAVE / 4; q++) {
        const int r = 4 * q + rsub;
        if (r >= rows_valid) continue;
        const double v0 = tile[r * TPITCH + 2 * csub], v1 = tile[r * TPITCH + 2 * csub + 1];
        double *dst = out_tile + (int64_t)r * pitch + 2 * csub;
        if (al16 && in0 && in1) {
            f64x2 v = {v0, v1};
            __builtin_nontemporal_store(v, reinterpret_cast<f64x2 *>(dst));
        } else {
            if (in0) dst[0] = v0;
            if (in1) dst[1] = v1;
        }
    }
}

// ---- TGLS: rolling sum with per-genotype error (garlic-roh.cpp:91-95)
// One wave per work item (run x 64 individuals).  Measured: grouping the 64-individual blocks of a
// run into one workgroup (shared L1 for the ncodes x 32 B term rows) was 15 % slower -- the kernel
// is bound by the three dependent memory rounds per tile at one wave per SIMD, not by L2 traffic.
constexpr int GL_WAVES = 1;
__global__ void __launch_bounds__(GL_WAVES * WAVE)
lod_chain_gl_kernel(VariantArgs p, int n_items)
{
    __shared__ double tiles[GL_WAVES][WAVE * TPITCH];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = blockIdx.x * GL_WAVES + wave;
    if (item >= n_items) return;
    double *tile = tiles[wave];
    const ChainItem it = p.items[item];
    if (it.chr < 0) return;
    const ChrDev c = p.chrs[it.chr];
    const int lane = threadIdx.x & (WAVE - 1), W = p.winsize, a = it.a, b = it.b;
    const int rows_valid = min(WAVE, p.ind_count - it.ind0);
    const int64_t col = (int64_t)p.ind_begin + it.ind0 + lane;
    const int64_t Gbase = c.loc_base + GOFF;
    double acc = 0.0;
    for (int l = a; l < a + W - 1; l++) acc += variant_term(p, Gbase + l, col); // first window, W-1 terms
    double *out_row0 = p.out + c.out_base + (int64_t)it.ind0 * c.out_pitch;
    // Per 32-window tile the 64 terms (32 entering, 32 leaving) are fetched in three rounds of
    // independent loads -- genotype words + dictionary codes, then the term gathers -- so a tile
    // costs three memory round trips instead of 64 x 2; pad rows keep every address in bounds,
    // masks are applied to the loaded values.
    for (int s0 = a & ~(TILE - 1); s0 <= b; s0 += TILE) {
        const int64_t Gin = Gbase + s0 + W - 1, Gout = Gbase + s0 - 1;
        uint32_t idx_in[TILE], idx_out[TILE];
#pragma unroll
        for (int j = 0; j < TILE; j++) {
            const uint32_t wi = p.packed[packed_index((Gin + j) >> 4, col, p.nwordrows)];
            const uint32_t wo = p.packed[packed_index((Gout + j) >> 4, col, p.nwordrows)];
            const uint32_t ci = p.codes[(Gin + j) * p.nind_pad + col];
            const uint32_t co = p.codes[(Gout + j) * p.nind_pad + col];
            idx_in[j] = ci * 4 + ((wi >> (2 * (int)((Gin + j) & 15))) & 3u);
            idx_out[j] = co * 4 + ((wo >> (2 * (int)((Gout + j) & 15))) & 3u);
        }
        double t_in[TILE], t_out[TILE];
#pragma unroll
        for (int j = 0; j < TILE; j++) {
            t_in[j] = p.tabgl[(Gin + j) * p.ncodes * 4 + idx_in[j]];
            t_out[j] = p.tabgl[(Gout + j) * p.ncodes * 4 + idx_out[j]];
        }
#pragma unroll
        for (int j = 0; j < TILE; j++) {
            const int s = s0 + j;
            const bool in = (s >= a && s <= b);
            const double ti = in ? t_in[j] : 0.0;
            const double to = (in && s > a) ? t_out[j] : 0.0;
            acc = (acc - to) + ti;
            tile[lane * TPITCH + j] = acc;
        }
        // the tile is this wave's own: order its LDS writes before the transposed reads
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        variant_store(tile, s0, a, b, lane, rows_valid, out_row0 + s0, c.out_pitch);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// Hand-off of LDS data between the waves of one workgroup through a counter in LDS.  A CU executes
// one wave's LDS instructions in issue order, so a counter written after the data is seen after the
// data and only the COMPILER must be kept from reordering: wavefront-scope fences.  (A
// workgroup-scope release also waits for the wave's global memory operations -- s_waitcnt vmcnt(0):
// the chain wave would wait for its prefetched next tile, the store wave for HBM to take the tile
// it has just issued, once per tile.)
// The counters themselves: relaxed workgroup-scope atomics on the __shared__ array itself, so that
// they stay ds_read / ds_write (through a `volatile int *` the compiler loses the address space and
// emits flat loads, which count as vector-memory operations too: every poll then waits vmcnt(0)).
#define LDS_FLAG_GET(x) __hip_atomic_load(&(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define LDS_FLAG_SET(x, v) __hip_atomic_store(&(x), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
__device__ __forceinline__ void lds_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); }
__device__ __forceinline__ void lds_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

// ---- TGLS ingest: per-genotype error probabilities -> one-byte dictionary codes on the device.
// dict_bits: the known values' bit patterns, sorted; dict_code: their codes.  A value that is not in
// the dictionary is reported (up to cap of them) and the host extends the dictionary and runs the
// rows again; after the first rows of a panel that practically never happens (GQ / PL integers).
constexpr int GL_DICT_MAX = 256;
__global__ void __launch_bounds__(256)
gl_encode_kernel(const double *__restrict__ gl, int64_t ld, int64_t locus_count, int32_t nind, int64_t nind_pad,
                 const uint64_t *__restrict__ dict_bits, const uint8_t *__restrict__ dict_code, int ndict,
                 uint8_t *__restrict__ codes, uint64_t *__restrict__ unknown, int32_t *__restrict__ n_unknown, int cap)
{
    __shared__ uint64_t bits_s[GL_DICT_MAX];
    __shared__ uint8_t code_s[GL_DICT_MAX];
    for (int k = threadIdx.x; k < ndict; k += blockDim.x) { bits_s[k] = dict_bits[k]; code_s[k] = dict_code[k]; }
    __syncthreads();
    const int64_t n = locus_count * nind;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t l = e / nind;
        const int i = (int)(e - l * nind);
        const uint64_t b = reinterpret_cast<const uint64_t *>(gl)[l * ld + i];
        int lo = 0, hi = ndict;                                 // first entry >= b
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (bits_s[mid] < b) lo = mid + 1; else hi = mid;
        }
        if (lo < ndict && bits_s[lo] == b) {
            codes[l * nind_pad + i] = code_s[lo];
        } else {
            const int k = atomicAdd(n_unknown, 1);
            if (k < cap) unknown[k] = b;
        }
    }
}

// Likelihoods that arrive dictionary-coded (garlic_panel_set_gl_codes): the caller's code -> the
// panel's code.  rows: [locus_count][ld] bytes.
__global__ void __launch_bounds__(256)
gl_recode_kernel(const uint8_t *__restrict__ rows, int64_t ld, int64_t locus_count, int32_t nind, int64_t nind_pad,
                 const uint8_t *__restrict__ remap, uint8_t *__restrict__ codes)
{
    __shared__ uint8_t map_s[256];
    map_s[threadIdx.x] = remap[threadIdx.x];
    __syncthreads();
    const int64_t n = locus_count * nind;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t l = e / nind;
        const int i = (int)(e - l * nind);
        codes[l * nind_pad + i] = map_s[rows[l * ld + i]];
    }
}

// ---- TGLS in two passes.  The term of (SNP, individual) does not depend on the window size, and
// looking it up costs two dependent loads plus a gather that drags in 15 cache lines of the
// ncodes x 32 B term row for 64 values.  Inside the sequential chain that latency is exposed three
// times per tile; as a pass of its own it is a fully parallel streaming kernel, done once per panel
// and reused by every window size.  terms: [64-individual block][GOFF + nloci + pad][64] doubles,
// pad rows 0.0 (= the term of a missing genotype).
__global__ void __launch_bounds__(256)
gl_terms_kernel(VariantArgs p, int64_t nloci, int64_t rows, double *__restrict__ terms)
{
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x >> 6;
    const int64_t blk = blockIdx.y, col = blk * WAVE + lane;
    const int64_t l0 = (int64_t)blockIdx.x * 64;
    for (int64_t l = l0 + wave; l < min(nloci, l0 + 64); l += 4) {
        const int64_t G = GOFF + l;
        const uint32_t word = p.packed[packed_index(G >> 4, col, p.nwordrows)];
        const uint32_t g = (word >> (2 * (int)(G & 15))) & 3u;
        const uint32_t code = p.codes[G * p.nind_pad + col];
        terms[(blk * rows + G) * WAVE + lane] = p.tabgl[((G * p.ncodes) + code) * 4 + g];
    }
}

// The same pass with the SNPs' table rows staged in LDS.  gl_terms_kernel gathers each term from the
// ncodes x 32 B row of its SNP in global memory -- up to 15 cache lines per wave and SNP, for every one of the
// panel's 64-individual blocks again (88.9 ms at 10M SNPs x 1250, where the 9.25 B per genotype it moves would
// take 22).  Here a workgroup owns GL_TERMS_S SNPs for ALL blocks: their rows come in once, coalesced, and the
// look-ups are LDS reads.  Dynamic LDS: GL_TERMS_S * ncodes * 4 doubles.
constexpr int GL_TERMS_S = 8;            // SNPs per workgroup: half a genotype word
__global__ void __launch_bounds__(256)
gl_terms_lds_kernel(VariantArgs p, int64_t nloci, int64_t rows, int nblk, const double *__restrict__ decay,
                    double *__restrict__ terms)
{
    extern __shared__ double gl_rows[];                      // [GL_TERMS_S][ncodes][4]
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x >> 6;
    const int64_t l0 = (int64_t)blockIdx.x * GL_TERMS_S;     // unpadded; GOFF is a multiple of 16, so l0 + GOFF is one of 8
    const int ns = (int)min<int64_t>(GL_TERMS_S, nloci - l0);
    const int64_t G0 = GOFF + l0;
    const int row_doubles = p.ncodes * 4;
    // decay != NULL: the scores of the weighted kernel straight away, (term * nomut) * norec as gl_scale_kernel makes
    // them (the same two multiplications, here on the table row: 4 * ncodes values instead of one per genotype)
    for (int e = threadIdx.x; e < ns * row_doubles; e += blockDim.x) {
        const double t = p.tabgl[G0 * row_doubles + e];
        const int64_t G = G0 + e / row_doubles;
        gl_rows[e] = decay ? (t * decay[2 * G]) * decay[2 * G + 1] : t;
    }
    __syncthreads();
    const int shift0 = 2 * (int)(G0 & 15);                   // 0 or 16: the chunk is one half of a genotype word
    for (int blk = wave; blk < nblk; blk += 4) {
        const int64_t col = (int64_t)blk * WAVE + lane;
        const uint32_t word = p.packed[packed_index(G0 >> 4, col, p.nwordrows)] >> shift0;
        uint32_t code[GL_TERMS_S];
#pragma unroll
        for (int u = 0; u < GL_TERMS_S; u++) code[u] = u < ns ? p.codes[(G0 + u) * p.nind_pad + col] : 0u;
#pragma unroll
        for (int u = 0; u < GL_TERMS_S; u++)
            if (u < ns)
                terms[((int64_t)blk * rows + G0 + u) * WAVE + lane] =
                    gl_rows[(u * p.ncodes + (int)code[u]) * 4 + (int)((word >> (2 * u)) & 3u)];
    }
}

// ---- Continuous likelihoods (--gl-type GL / PL: more distinct values than a dictionary holds).
// The error probabilities live in the term matrix's layout, vals[blk][rows][64]; the terms come from
// lod() evaluated on the device with glibc's log10 restated (tgls_math.hpp).
// gl_store_kernel: caller rows [locus_count][ld] -> vals (loci [l0, l0 + locus_count)).
__global__ void __launch_bounds__(256)
gl_store_kernel(const double *__restrict__ gl, int64_t ld, int64_t l0, int64_t locus_count, int32_t nind,
                int64_t rows, double *__restrict__ vals)
{
    const int64_t n = locus_count * nind;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t l = e / nind;
        const int64_t i = e - l * nind;
        vals[((i >> 6) * rows + GOFF + l0 + l) * WAVE + (i & 63)] = gl[l * ld + i];
    }
}

// the dictionary overflowed: what has been coded so far becomes values (codes: [rows][nind_pad])
__global__ void __launch_bounds__(256)
gl_decode_kernel(const uint8_t *__restrict__ codes, const double *__restrict__ dict, int64_t nind_pad, int64_t rows,
                 double *__restrict__ vals)
{
    __shared__ double d_s[GL_DICT_MAX];
    d_s[threadIdx.x] = dict[threadIdx.x];
    __syncthreads();
    const int64_t n = rows * nind_pad;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t G = e / nind_pad;
        const int64_t i = e - G * nind_pad;
        vals[((i >> 6) * rows + G) * WAVE + (i & 63)] = d_s[codes[e]];
    }
}

// caller's one-byte codes + value table straight to values (garlic_panel_set_gl_codes in continuous mode)
__global__ void __launch_bounds__(256)
gl_store_codes_kernel(const uint8_t *__restrict__ rows_in, int64_t ld, int64_t l0, int64_t locus_count, int32_t nind,
                      const double *__restrict__ dict, int64_t rows, double *__restrict__ vals)
{
    __shared__ double d_s[GL_DICT_MAX];
    d_s[threadIdx.x] = dict[threadIdx.x];
    __syncthreads();
    const int64_t n = locus_count * nind;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t l = e / nind;
        const int64_t i = e - l * nind;
        vals[((i >> 6) * rows + GOFF + l0 + l) * WAVE + (i & 63)] = d_s[rows_in[l * ld + i]];
    }
}

// terms[blk][G][lane] = lod(genotype, freq[G], vals[blk][G][lane]) for the padded rows [G0, G1); vals and
// terms may be the same buffer (each element is read, then written, by one thread).  freq: [rows],
// pad rows hold 0 (-> term +0.0, like the code-3 genotypes of pad rows and pad columns).
// min_bits (GL_MIN_SLOTS entries, zeroed by the caller; may be NULL): the most negative finite term, for
// lod_exact_needed -- among negative doubles the most negative has the largest bit pattern, so an unsigned
// atomicMax per workgroup, spread over the slots, collects it (a reduction pass of its own read the matrix again:
// 28 ms at 10M SNPs x 1250).
constexpr int GL_MIN_SLOTS = 1024;
__global__ void __launch_bounds__(256)
gl_terms_cont_kernel(const uint32_t *__restrict__ packed, int64_t nwordrows, const double *__restrict__ freq,
                     const double *__restrict__ logtab, const double *vals, int64_t G0, int64_t G1, int64_t rows,
                     double *terms, unsigned long long *__restrict__ min_bits)
{
    __shared__ double tab_s[256];
    __shared__ unsigned long long red[4];
    tab_s[threadIdx.x] = logtab[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x >> 6;
    const int64_t blk = blockIdx.y, col = blk * WAVE + lane;
    const int64_t g0 = G0 + (int64_t)blockIdx.x * 64;
    double m = 0.0;
    for (int64_t G = g0 + wave; G < min(G1, g0 + 64); G += 4) {
        const uint32_t word = packed[packed_index(G >> 4, col, nwordrows)];
        const uint32_t g = (word >> (2 * (int)(G & 15))) & 3u;
        const int64_t at = (blk * rows + G) * WAVE + lane;
        const double t = lod_term(g, freq[G], vals[at], tab_s);
        terms[at] = t;
        if (t < m && t > -1.7976931348623157e308) m = t;     // NaN and -inf fail the comparisons
    }
    if (min_bits) {
        unsigned long long b = m < 0.0 ? f64_bits(m) : 0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) b = max(b, (unsigned long long)__shfl_xor((long long)b, o));
        if (lane == 0) red[wave] = b;
        __syncthreads();
        if (threadIdx.x == 0) {
            b = max(max(red[0], red[1]), max(red[2], red[3]));
            if (b) atomicMax(min_bits + ((blockIdx.x * 7u + blockIdx.y * 131u) & (unsigned)(GL_MIN_SLOTS - 1)), b);
        }
    }
}

// start-up check of the device's log10 against the host's: out[i] = glibc_log10(in[i])
__global__ void __launch_bounds__(256)
log10_probe_kernel(const double *__restrict__ in, const double *__restrict__ logtab, int64_t n, double *__restrict__ out)
{
    __shared__ double tab_s[256];
    tab_s[threadIdx.x] = logtab[threadIdx.x];
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = glibc_log10(in[i], tab_s);
}

// wLOD with per-genotype likelihoods reads scores, (term * nomut) * norec (garlic-roh.cpp:249), from
// the same matrix: scaled in place (and rebuilt by gl_terms_kernel when the unweighted TGLS chain
// needs the raw terms again -- a session normally uses one of the two).  decay: [rows][2].
__global__ void __launch_bounds__(256)
gl_scale_kernel(double *__restrict__ terms, const double *__restrict__ decay, int64_t rows, int64_t nblk)
{
    const int64_t n = rows * nblk * WAVE;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t G = (i / WAVE) % rows;
        terms[i] = (terms[i] * decay[2 * G]) * decay[2 * G + 1];
    }
}

// The chain of lod_chain_gl_kernel with its 64 terms per tile read straight from the term matrix
// (one round of coalesced 512-B loads per tile).  All work items are resident at once, so the
// kernel lasts as long as the longest run needs for its tiles one after the other; that path is
// split over two waves: wave 0 loads (one tile ahead) and runs the chain into one of two LDS
// tiles, wave 1 writes finished tiles out (transposed, 16-B non-temporal stores).  Two counters in
// LDS (tiles written / tiles stored) instead of barriers, as in the unweighted kernel.
// Every term is read twice (entering the window and, W-1 SNPs later, leaving it); at scale the
// second read misses L2 (FETCH_SIZE x 2 = 40 GB for 20 GB of terms at 2M x 1280) but not the
// memory-side cache: a variant that staged the rows through a 124-KB LDS ring (LDS-DMA, one fetch
// per term, one workgroup per CU) ran no faster (9.4 vs 8.8 ms) -- terms once + scores once at
// 16 B per window is what the kernel moves through HBM either way, at the rate the unweighted
// kernel writes at on the same box.
__global__ void __launch_bounds__(2 * WAVE)
lod_chain_terms_kernel(VariantArgs p, int n_items, int64_t rows, const double *__restrict__ terms)
{
    __shared__ double tiles[2][WAVE * TPITCH];
    __shared__ int flags[2];                               // [0] tiles written, [1] tiles stored
    const int item = blockIdx.x;
    if (item >= n_items) return;
    const ChainItem it = p.items[item];
    if (it.chr < 0) return;
    const ChrDev c = p.chrs[it.chr];
    const int lane = threadIdx.x & (WAVE - 1), W = p.winsize, a = it.a, b = it.b;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rows_valid = min(WAVE, p.ind_count - it.ind0);
    double *out_row0 = p.out + c.out_base + (int64_t)it.ind0 * c.out_pitch;
    if (threadIdx.x == 0) { flags[0] = 0; flags[1] = 0; }
    __syncthreads();
    const int first = a & ~(TILE - 1);

    if (wave == 1) {   // ---- write-out
        int k = 0;
        for (int s0 = first; s0 <= b; s0 += TILE, k++) {
            while (LDS_FLAG_GET(flags[0]) <= k) __builtin_amdgcn_s_sleep(1);
            lds_acquire();
            variant_store(tiles[k & 1], s0, a, b, lane, rows_valid, out_row0 + s0, c.out_pitch);
            lds_release();
            if (lane == 0) LDS_FLAG_SET(flags[1], k + 1);
        }
        return;
    }

    // ---- loads + chain
    const int64_t col = (int64_t)p.ind_begin + it.ind0 + lane;
    const double *tcol = terms + ((col >> 6) * rows) * WAVE + (col & 63);   // term of SNP G at tcol[G * 64]
    const int64_t Gbase = c.loc_base + GOFF;
    double acc = 0.0;
    for (int l0 = a; l0 < a + W - 1; l0 += 32) {         // first window, W-1 terms, 32 loads in flight
        double t[32];
#pragma unroll
        for (int q = 0; q < 32; q++) t[q] = tcol[(Gbase + min(l0 + q, a + W - 2)) * WAVE];
#pragma unroll
        for (int q = 0; q < 32; q++) acc += (l0 + q < a + W - 1) ? t[q] : 0.0;
    }
    double n_in[TILE], n_out[TILE];                       // the next tile's terms (pad rows keep it in bounds)
    {
        const int64_t Gin = Gbase + first + W - 1, Gout = Gbase + first - 1;
#pragma unroll
        for (int j = 0; j < TILE; j++) {
            n_in[j] = tcol[(Gin + j) * WAVE];
            n_out[j] = tcol[(Gout + j) * WAVE];
        }
    }
    int k = 0;
    for (int s0 = first; s0 <= b; s0 += TILE, k++) {
        const int64_t Gin = Gbase + s0 + W - 1, Gout = Gbase + s0 - 1;
        double t_in[TILE], t_out[TILE];
#pragma unroll
        for (int j = 0; j < TILE; j++) {
            t_in[j] = n_in[j];
            t_out[j] = n_out[j];
        }
#pragma unroll
        for (int j = 0; j < TILE; j++) {
            n_in[j] = tcol[(Gin + TILE + j) * WAVE];
            n_out[j] = tcol[(Gout + TILE + j) * WAVE];
        }
        while (LDS_FLAG_GET(flags[1]) + 2 <= k) __builtin_amdgcn_s_sleep(1);   // tile buffer k & 1 has been written out
        lds_acquire();
        double *tile = tiles[k & 1];
#pragma unroll
        for (int j = 0; j < TILE; j++) {
            const int s = s0 + j;
            const bool in = (s >= a && s <= b);
            const double ti = in ? t_in[j] : 0.0;
            const double to = (in && s > a) ? t_out[j] : 0.0;
            acc = (acc - to) + ti;
            tile[lane * TPITCH + j] = acc;
        }
        lds_release();
        if (lane == 0) LDS_FLAG_SET(flags[0], k + 1);
    }
}

// 1.0 / x with a NaN handed on as x86 does (the operand, quieted): nothing may depend on how the
// division expansion of gfx950 treats payloads
__device__ __forceinline__ double reciprocal_x86(double x)
{
    return x != x ? f64_from_bits(f64_bits(x) | 0x0008000000000000ull) : 1.0 / x;
}

// ---- 1.0 / LD, IEEE division (garlic-roh.cpp:270 does it per use; the quotient is the same double)
__global__ void reciprocal_kernel(const double *__restrict__ ld, double *__restrict__ rld, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) rld[i] = reciprocal_x86(ld[i]);
}

// ---- The reference's rolling sum to the letter, for the inputs where the letter matters.
// calcLOD decides "the previous window holds no score" by VALUE (garlic-roh.cpp:79: win[ind][locus-1] !=
// MISSING): a scored window whose sum is exactly -9999.0 makes the next window a fresh left-to-right sum of
// its W terms instead of (previous - leaving) + entering -- the same real number, another rounding.  The
// tuned chains carry "previous window scored" by position.  The two can only differ when a window sum can
// reach -9999 at all: the host checks W * (most negative term of the panel) against that (lod_exact_needed
// in garlic_hip.hip; with --error 0.001 a window would need 3333 heterozygous SNPs), and only then runs this
// kernel: one wavefront per (run, 64 individuals), lane = individual, every term looked up in memory, the
// reference's control flow per window.  Slow (it is never on a realistic input's path), exact.
__global__ void __launch_bounds__(WAVE)
lod_chain_exact_kernel(VariantArgs p, int n_items)
{
    __shared__ double tile[WAVE * TPITCH];
    const int item = blockIdx.x;
    if (item >= n_items) return;
    const ChainItem it = p.items[item];
    if (it.chr < 0) return;
    const ChrDev c = p.chrs[it.chr];
    const int lane = threadIdx.x, W = p.winsize, a = it.a, b = it.b;
    const int rows_valid = min(WAVE, p.ind_count - it.ind0);
    const int64_t col = (int64_t)p.ind_begin + it.ind0 + lane;
    const int64_t Gbase = c.loc_base + GOFF;
    double *out_row0 = p.out + c.out_base + (int64_t)it.ind0 * c.out_pitch;
    double prev = MISSING_D;                               // the window in front of a run holds no score
    for (int s0 = a & ~(TILE - 1); s0 <= b; s0 += TILE) {
        for (int j = 0; j < TILE; j++) {
            const int s = s0 + j;
            double v = 0.0;
            if (s >= a && s <= b) {
                if (prev != MISSING_D) {                   // garlic-roh.cpp:79, 92-100 (NaN != MISSING: rolls on)
                    v = (prev - variant_term(p, Gbase + s - 1, col)) + variant_term(p, Gbase + s + W - 1, col);
                } else {                                   // :57-71 / :106-120: from 0, left to right
                    for (int i = 0; i < W; i++) v += variant_term(p, Gbase + s + i, col);
                }
                prev = v;
            }
            tile[lane * TPITCH + j] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        variant_store(tile, s0, a, b, lane, rows_valid, out_row0 + s0, c.out_pitch);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// most negative finite value of a double array (one partial per workgroup; the host takes their minimum)
__global__ void __launch_bounds__(256)
min_finite_kernel(const double *__restrict__ x, int64_t n, double *__restrict__ partial)
{
    __shared__ double red[256];
    double m = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = x[i];
        if (v < m && v > -1.7976931348623157e308) m = v;   // NaN and -inf fail the comparisons
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) red[threadIdx.x] = fmin(red[threadIdx.x], red[threadIdx.x + d]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// ---- wLOD: every valid window summed afresh (garlic-roh.cpp:253-273)
// dynamic LDS: score ring [ring][64] doubles | rld row [winsize] | transpose tile [64][TPITCH]
__global__ void __launch_bounds__(WAVE)
wlod_kernel(VariantArgs p, int ring)
{
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    double *score = dyn;                              // [ring][64], slot = locus % ring
    double *rrow = dyn + (size_t)ring * WAVE;         // [winsize]
    double *tile = rrow + ((p.winsize + 1) & ~1);     // [64][TPITCH]
    const ChainItem it = p.items[blockIdx.x];
    if (it.chr < 0) return;
    const ChrDev c = p.chrs[it.chr];
    const int lane = threadIdx.x, W = p.winsize, a = it.a, b = it.b;
    const int rows_valid = min(WAVE, p.ind_count - it.ind0);
    const int64_t col = (int64_t)p.ind_begin + it.ind0 + lane;
    const int64_t Gbase = c.loc_base + GOFF;
    double *out_row0 = p.out + c.out_base + (int64_t)it.ind0 * c.out_pitch;
    int have = a; // scores of loci [a, have) are in the ring
    for (int s0 = a & ~(TILE - 1); s0 <= b; s0 += TILE) {
        const int need = min(b, s0 + TILE - 1) + W; // loci < need are used by this tile
        for (int l = have; l < need; l++) {
            const int64_t G = Gbase + l;
            // (lod * nomut) * norec, in that order (garlic-roh.cpp:249)
            score[(l % ring) * WAVE + lane] = (variant_term(p, G, col) * p.decay[2 * G]) * p.decay[2 * G + 1];
        }
        have = max(have, need);
        for (int j = 0; j < TILE; j++) {
            const int s = s0 + j;
            double sum = 0.0;
            if (s >= a && s <= b) {
                const double *src = p.rld + (c.loc_base + s) * (int64_t)W;
                __syncthreads();
                for (int k = lane; k < W; k += WAVE) rrow[k] = src[k];
                __syncthreads();
                for (int k = 0; k < W; k++)
                    sum += score[((s + k) % ring) * WAVE + lane] * rrow[k]; // product rounded, then added
            }
            tile[lane * TPITCH + j] = sum;
        }
        __syncthreads();
        variant_store(tile, s0, a, b, lane, rows_valid, out_row0 + s0, c.out_pitch);
        __syncthreads();
    }
}

// ---- wLOD, tuned path (no per-genotype likelihoods, winsize >= R).
// Every window is an independent ordered sum, so the work is FP64-VALU bound: 2 instructions
// (v_mul_f64, v_add_f64 -- the product is rounded before the add, garlic-roh.cpp:262-268) per
// (window, term).  Lane = individual; a wave keeps R window accumulators per lane in registers and
// walks the SNPs l once: the lane's score sc[l] (one LDS look-up of the per-SNP row
// wtab[l][genotype] = (lod * nomut) * norec, tabulated on the host with the reference's operation
// order) feeds the R windows that contain l, each with its own weight.  The weights of one SNP
// for R consecutive windows are contiguous in the skewed table D[l][j] = 1/LD[l-j][j] and
// wave-uniform, so they arrive through the scalar cache and cost no vector instruction.
//
//   grid.x = 32-window tiles of all chromosomes, grid.y = 64-individual blocks; the tile is
//   transposed through LDS so that every store covers whole row segments, and windows that hold
//   no score (mask byte 0) are written as MISSING by the same store: no separate fill pass.
constexpr int WLOD_R = 16;   // window accumulators per lane (weights of one step: 32 SGPRs)
constexpr int WLOD_WAVES = 4;  // waves (64-individual blocks) per workgroup (8 was measured slower)
constexpr int WT_PITCH = 18;   // doubles per row of the write-out patch (16 + pad, 16-B aligned rows)
constexpr int SKEW_FRONT = 16; // doubles of padding in front of the skewed weight table

// coverage bits instead of scores (garlic_roh_coverage_fused with --weighted): every group leaves 16 bits per
// individual -- score >= cutoff, MISSING and NaN never -- in a bit matrix [chromosome][individual][locus / 32] that
// cov_counts_from_bits_kernel (coverage_kernel.hpp) turns into inWin[]; bits == NULL: scores as usual
struct CovBits {
    uint32_t *bits;
    const ChrDev *bchrs;       // out_base / out_pitch in dwords, per chromosome
    double cutoff;
};

struct WlodArgs {
    const uint8_t *valid;      // [nloci] 1 = window holds a score
    const ChrDev *chrs;
    const int2 *tiles;         // per 32-window tile: {chromosome, first window (chromosome-local)}
    int64_t nwordrows;
    int32_t nchr, ind_begin, ind_count, winsize, nquad;   // nquad = workgroups per tile
    uint32_t n_work;           // tiles x nquad
    int32_t use_patch;         // transposed write-out through the LDS patch (allocated then)
    int64_t score_rows;        // FROM_SCORES: SNP rows per 64-individual block of the term matrix
    int32_t gl_ring;           // FROM_SCORES: hand-scheduled loop with per-wave LDS rings of term rows (allocated then)
    CovBits cov;
    // a launch that repairs another one: runs only if *run_if != 0 (the strip kernel's "a wave ran out of its poll budget"
    // flag, wlod_strip_kernel.hpp) and then counts itself in *rerun_count -- enqueued behind every strip launch, so that
    // no call has to come back to the host to look at the flag; NULL: an ordinary launch
    const int32_t *run_if;
    int32_t *rerun_count;
};
// dynamic LDS of the term-matrix variant: patch lock (16 B) + patch [64][WT_PITCH] doubles, then, 1-KB
// aligned, one ring of GARLIC_WLOD_GL_RING_ROWS x 512 B per wave
constexpr uint32_t WLOD_GL_RING_OFF = (16u + 64u * 18u * 8u + 1023u) & ~1023u;

// Ordered sums of windows s .. s+15 for this lane's individual: acc[r] = sum_j sc[s+r+j] * D[s+r+j][j],
// j ascending from +0.0 (garlic-roh.cpp:255-272).  The whole loop is the hand-scheduled block of
// wlod_loop_gfx950.inc (tools/gen_wlod_asm.py); this wrapper only prepares its operands.
//   rows  LDS score rows, row i = SNP s+i         gcol  this lane's genotype column (word row w at gcol[w*64])
//   G     padded global index of SNP s             Ds    D + (unpadded global index of SNP s) * W
template <int R>
__device__ __forceinline__ void wlod_group(const double *rows, const uint32_t *gcol, int64_t G,
                                           const double *Ds, int W, double (&acc)[R])
{
    static_assert(R == 16, "the hand-scheduled loop keeps 16 weights per step in SGPRs");
    // genotype words of this lane, one word (16 SNPs) of look-ahead; gaddr = next row to fetch
    uint64_t gaddr = reinterpret_cast<uint64_t>(gcol + (G >> 4) * WAVE);
    uint32_t bit = 2 * (uint32_t)(G & 15);
    double sc, scn, t0, t1;
    uint32_t vt, word, nextw;
    uint32_t n = (uint32_t)(W - (R - 1));        // steps in which all 16 windows are active
    uint32_t row = (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) double *)rows);
    const double *dp = Ds - (R - 1);             // step 0: elements 15-r of {D[s][-15] .. D[s][0]}
    const uint32_t stride = (uint32_t)(W + 1) * 8u;
    asm volatile(GARLIC_WLOD_LOOP_ASM
                 : [a0] "=&v"(acc[0]), [a1] "=&v"(acc[1]), [a2] "=&v"(acc[2]), [a3] "=&v"(acc[3]),
                   [a4] "=&v"(acc[4]), [a5] "=&v"(acc[5]), [a6] "=&v"(acc[6]), [a7] "=&v"(acc[7]),
                   [a8] "=&v"(acc[8]), [a9] "=&v"(acc[9]), [a10] "=&v"(acc[10]), [a11] "=&v"(acc[11]),
                   [a12] "=&v"(acc[12]), [a13] "=&v"(acc[13]), [a14] "=&v"(acc[14]), [a15] "=&v"(acc[15]),
                   [sc] "=&v"(sc), [scn] "=&v"(scn), [t0] "=&v"(t0), [t1] "=&v"(t1), [vt] "=&v"(vt),
                   [word] "=&v"(word), [nextw] "=&v"(nextw), [gaddr] "+v"(gaddr), [bit] "+s"(bit),
                   [row] "+s"(row), [n] "+s"(n)
                 : [dp] "s"(dp), [stride] "s"(stride), [rowbytes] "s"((uint64_t)(WAVE * 4))
                 : GARLIC_WLOD_LOOP_CLOBBERS);
}

// The same loop for per-genotype likelihoods (GARLIC_WLOD_GL_LOOP_ASM): the score of (SNP, lane) is the
// lane's entry of the scaled TGLS term matrix.  The wave stages the block's 512-B rows through its
// own LDS ring (LDS-DMA, two rows per request, GARLIC_WLOD_GL_RING_ROWS - 2 rows ahead) and reads
// them with ds_read_b64 at row + lane * 8 where the plain loop looks a genotype up.
//   ring_lds  LDS byte address of this wave's ring (1-KB aligned)
//   trow      the block's row of SNP s in the term matrix (wave-uniform; 64 doubles per row)
template <int R>
__device__ __forceinline__ void wlod_group_gl(uint32_t ring_lds, const double *trow, int lane, const double *Ds, int W,
                                              double (&acc)[R])
{
    static_assert(R == 16, "the hand-scheduled loop keeps 16 weights per step in SGPRs");
    double sc, scn, t0, t1;
    uint32_t vt, voff16 = (uint32_t)lane * 16u, rd = 0, wr = 0;
    uint32_t n = (uint32_t)(W - (R - 1));
    const uint32_t lane8b = ring_lds + (uint32_t)lane * 8u;
    const double *dp = Ds - (R - 1);
    const uint32_t stride = (uint32_t)(W + 1) * 8u;
    asm volatile(GARLIC_WLOD_GL_LOOP_ASM
                 : [a0] "=&v"(acc[0]), [a1] "=&v"(acc[1]), [a2] "=&v"(acc[2]), [a3] "=&v"(acc[3]),
                   [a4] "=&v"(acc[4]), [a5] "=&v"(acc[5]), [a6] "=&v"(acc[6]), [a7] "=&v"(acc[7]),
                   [a8] "=&v"(acc[8]), [a9] "=&v"(acc[9]), [a10] "=&v"(acc[10]), [a11] "=&v"(acc[11]),
                   [a12] "=&v"(acc[12]), [a13] "=&v"(acc[13]), [a14] "=&v"(acc[14]), [a15] "=&v"(acc[15]),
                   [sc] "=&v"(sc), [scn] "=&v"(scn), [t0] "=&v"(t0), [t1] "=&v"(t1), [vt] "=&v"(vt),
                   [voff16] "+v"(voff16), [rd] "+s"(rd), [wr] "+s"(wr), [n] "+s"(n)
                 : [dp] "s"(dp), [stride] "s"(stride), [lane8b] "v"(lane8b), [trow] "s"(trow), [rbase] "s"(ring_lds)
                 : GARLIC_WLOD_LOOP_CLOBBERS);
}

// wLOD with per-genotype likelihoods (garlic-roh.cpp:245, USE_GL): the term of (SNP, individual)
// comes from the TGLS term matrix -- one coalesced 512-B load per step instead of an LDS look-up --
// and is scaled by the SNP's two decay factors (wave-uniform, scalar loads) on the fly.  This
// variant is compiler-scheduled (terms are fetched 8 steps at a time, the weights by the scalar
// loads hipcc places); the hand-scheduled loop above assumes the LDS look-up.
typedef const __attribute__((address_space(4))) double *const_f64_ptr;

template <int R>
__device__ __forceinline__ void wlod_group_scores(const double *tcol, int64_t G, const double *Ds, int W,
                                                  double (&acc)[R])
{
    // weights through the constant address space: they never change during the kernel, and only
    // then does the compiler keep their wave-uniform loads on the scalar path
    const const_f64_ptr Dg = (const_f64_ptr)(uintptr_t)Ds;
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = 0.0;
    const double *tp = tcol + G * WAVE;
    // score of SNP s+i = (lod * nomut) * norec (garlic-roh.cpp:249): the term matrix has been scaled
    // in place by gl_scale_kernel (scaling on the fly from two scalar loads per SNP was 30 % slower,
    // a second matrix of scores costs another 8 B per genotype)
    auto score = [&](int i) -> double { return tp[i * WAVE]; };
    double up[R - 1];
#pragma unroll
    for (int i = 0; i < R - 1; i++) up[i] = score(i);
#pragma unroll
    for (int i = 0; i < R - 1; i++) {                      // windows enter one by one
        const const_f64_ptr Dr = Dg + (int64_t)i * W;
#pragma unroll
        for (int r = 0; r <= i; r++) acc[r] += up[i] * Dr[i - r];
    }
    int i = R - 1;
    for (; i + 8 <= W; i += 8) {                           // all R windows take every SNP
        double sc[8];
#pragma unroll
        for (int q = 0; q < 8; q++) sc[q] = score(i + q);
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const const_f64_ptr Dr = Dg + (int64_t)(i + q) * W + (i + q - (R - 1));
#pragma unroll
            for (int r = 0; r < R; r++) acc[r] += sc[q] * Dr[(R - 1) - r];
        }
    }
    for (; i < W; i++) {
        const double sc = score(i);
        const const_f64_ptr Dr = Dg + (int64_t)i * W + (i - (R - 1));
#pragma unroll
        for (int r = 0; r < R; r++) acc[r] += sc * Dr[(R - 1) - r];
    }
    double dn[R - 1];
#pragma unroll
    for (int d = 0; d < R - 1; d++) dn[d] = score(W + d);
#pragma unroll
    for (int d = 0; d < R - 1; d++) {                      // and leave one by one
        const const_f64_ptr Dr = Dg + (int64_t)(W + d) * W;
#pragma unroll
        for (int r = d + 1; r < R; r++) acc[r] += dn[d] * Dr[W + d - r];
    }
}

// Windows narrower than the group (W < R: GARLIC's default --winsize is 10): no step has all R windows active,
// so the hand-scheduled loops above (R-1 steps of windows entering, W-(R-1) full steps, R-1 leaving) do not apply.
// Step i (SNP s+i, i = 0 .. W+R-2) serves the windows r with 0 <= i-r < W, weight D[s+i][i-r]; with W a template
// argument everything is unrolled and the compiler schedules it.  These shapes are bound by their output (8 B
// per window against 2 W flops), not by the loop; the generic one-wave-per-run kernel they replace took 151 ms
// for 2M SNPs x 1280 individuals at W = 15, the tile kernel 7.5 ms at W = 16.
template <int R, int WC, class ScoreFn>
__device__ __forceinline__ void wlod_group_small_w(ScoreFn score, const double *Ds, double (&acc)[R])
{
    const const_f64_ptr Dg = (const_f64_ptr)(uintptr_t)Ds;      // wave-uniform weights: scalar loads
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = 0.0;
    double sc[WC + R - 1];
#pragma unroll
    for (int i = 0; i < WC + R - 1; i++) sc[i] = score(i);
#pragma unroll
    for (int i = 0; i < WC + R - 1; i++) {
        const const_f64_ptr Dr = Dg + (int64_t)i * WC;
#pragma unroll
        for (int r = 0; r < R; r++)
            if (i - r >= 0 && i - r < WC) acc[r] += sc[i] * Dr[i - r];      // ascending j = i - r per window, from +0.0
    }
}

template <int R, class ScoreFn>
__device__ __forceinline__ void wlod_group_small(ScoreFn score, const double *Ds, int W, double (&acc)[R])
{
    static_assert(R == 16, "one case per window size below the group size");
    switch (W) {
    case 2: wlod_group_small_w<R, 2>(score, Ds, acc); break;
    case 3: wlod_group_small_w<R, 3>(score, Ds, acc); break;
    case 4: wlod_group_small_w<R, 4>(score, Ds, acc); break;
    case 5: wlod_group_small_w<R, 5>(score, Ds, acc); break;
    case 6: wlod_group_small_w<R, 6>(score, Ds, acc); break;
    case 7: wlod_group_small_w<R, 7>(score, Ds, acc); break;
    case 8: wlod_group_small_w<R, 8>(score, Ds, acc); break;
    case 9: wlod_group_small_w<R, 9>(score, Ds, acc); break;
    case 10: wlod_group_small_w<R, 10>(score, Ds, acc); break;
    case 11: wlod_group_small_w<R, 11>(score, Ds, acc); break;
    case 12: wlod_group_small_w<R, 12>(score, Ds, acc); break;
    case 13: wlod_group_small_w<R, 13>(score, Ds, acc); break;
    case 14: wlod_group_small_w<R, 14>(score, Ds, acc); break;
    default: wlod_group_small_w<R, 15>(score, Ds, acc); break;
    }
}

// Write-out of one group of R windows x 64 individuals (accumulators in registers): windows without a
// score become MISSING (garlic-roh.cpp:232); then 128 contiguous bytes per row and instruction,
// non-temporal, transposed through ONE LDS patch [64][WT_PITCH] per workgroup that its waves take turns
// on (a patch per wave would cost the occupancy the scalar weight loads need; 16 B per lane and row
// straight from registers left 0.45 ms of L2 write-back work per 1.6 GB).  The host enables the patch
// while score rows + patch still allow 8 waves per SIMD; otherwise (wide windows, dense / unaligned
// layouts) each lane writes its own row.
template <int R, bool ALIGNED16, class Args>
__device__ __forceinline__ void wlod_write_group(double (&acc)[R], uint32_t gm, const ChrDev &c, const Args &p,
                                                 double *__restrict__ out, double *patch, int *patch_lock,
                                                 int ind0, int s0, int grp, int lane_hint, int chr)
{
    // the lane index afresh (opaque to the compiler): everything per-lane below is then computed here instead of being
    // kept alive across the hand-scheduled loop, whose register budget leaves no room -- hipcc spilled those values and
    // reloaded them in here behind s_waitcnt vmcnt(0), i.e. behind the previous block's score stores
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    (void)lane_hint;
    const bool row_ok = ind0 + lane < p.ind_count;
    if (p.cov.bits) {          // 2 bytes per lane and group instead of 128
        uint32_t m = 0;
#pragma unroll
        for (int r = 0; r < R; r++) m |= (((gm >> r) & 1u) && acc[r] >= p.cov.cutoff) ? (1u << r) : 0u;     // NaN >= x is false
        if (row_ok) {
            const ChrDev bc = p.cov.bchrs[chr];
            uint16_t *brow = reinterpret_cast<uint16_t *>(p.cov.bits + bc.out_base + (int64_t)(ind0 + lane) * bc.out_pitch);
            brow[(s0 + grp * R) >> 4] = (uint16_t)m;
        }
        return;
    }
#ifdef GARLIC_WLOD_ABL_NO_WRITE        // timing experiment: only one value per lane leaves (results wrong)
    if (acc[0] == 1.2345e-300 && row_ok) out[ind0 + lane] = acc[0];
    return;
#endif
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = (gm != 0 && ((gm >> r) & 1u)) ? acc[r] : MISSING_D;
    const int sg = s0 + grp * R;
    if (ALIGNED16 && (p.use_patch & 1)) {
        // The lock guards LDS only, and a CU serves the LDS operations of its waves in the order they arrive: the
        // holder's patch reads are issued before its unlock, the next holder's writes after its successful CAS.  So the
        // fences are wavefront-scope (compiler ordering only).  Workgroup-scope acquire / release also wait for the
        // wave's GLOBAL operations (s_waitcnt vmcnt(0)): every write-out then waited until memory had taken the 16
        // stores it had just issued -- 12 % of the kernel at W = 100 (tools/exp/wlod_abl2.sh: 18.1 ms, 15.9 without
        // the write-out).
        if (lane == 0)
            while (atomicCAS(patch_lock, 0, 1) != 0) __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < R; r += 2)
            *reinterpret_cast<double2 *>(patch + lane * WT_PITCH + r) = make_double2(acc[r], acc[r + 1]);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int cc = 2 * (lane & 7);
        double *out_blk = out + c.out_base + (int64_t)ind0 * c.out_pitch + s0 + grp * R + cc;
        const double *prow = patch + (lane >> 3) * WT_PITCH + cc;
        double *dst = out_blk + (int64_t)(lane >> 3) * c.out_pitch;
#pragma unroll 1
        for (int q = 0; q < 8; q++, prow += 8 * WT_PITCH, dst += 8 * c.out_pitch) {
            // (one row piece at a time, rolled: registers are what buys occupancy here)
            const int rrow = 8 * q + (lane >> 3);
            const double2 v = *reinterpret_cast<const double2 *>(prow);
            if (ind0 + rrow >= p.ind_count) continue;
            if (sg + cc + 1 < c.nloci) {
                __builtin_nontemporal_store(v.x, dst);
                __builtin_nontemporal_store(v.y, dst + 1);
            } else if (sg + cc < c.nloci) {
                dst[0] = v.x;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) atomicExch(patch_lock, 0);
    } else if (row_ok) {
        double *out_row = out + c.out_base + (int64_t)(ind0 + lane) * c.out_pitch + s0;
#pragma unroll
        for (int r = 0; r < R; r += 2) {
            if (ALIGNED16 && sg + r + 1 < c.nloci) {
                *reinterpret_cast<double2 *>(out_row + grp * R + r) = make_double2(acc[r], acc[r + 1]);
            } else {
                if (sg + r < c.nloci) out_row[grp * R + r] = acc[r];
                if (sg + r + 1 < c.nloci) out_row[grp * R + r + 1] = acc[r + 1];
            }
        }
    }
}

template <int R, bool ALIGNED16, bool FROM_SCORES, bool GL_RING = false, bool SMALLW = false>
__device__ __forceinline__ void
wlod_tile_body(const uint32_t *__restrict__ packed,
               const double *__restrict__ wtab,   // [GOFF + nloci + pad][4]; FROM_SCORES: term matrix [blk][rows][64]
               const double *__restrict__ D,      // [nloci + pad][W], D[l][j] = 1.0 / LD[l - j][j]
               double *__restrict__ out, const WlodArgs &p)
{   // the read-only tables are separate __restrict__ arguments: only then are the wave-uniform
    // weight loads provably unclobbered by the score stores and issued as scalar loads
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    // workgroup = WLOD_WAVES waves = that many 64-individual blocks of ONE tile: they share the
    // staged score rows (LDS per wave stays small enough for 8 waves per SIMD at any W) and
    // walk the same weights at the same time (scalar-cache hits for all but the first)
    const int lane = threadIdx.x & (WAVE - 1), W = p.winsize;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *rows = dyn;                                   // [W + TILE][4]   (not with FROM_SCORES)
    const size_t rows_doubles = FROM_SCORES ? 0 : (size_t)((W + TILE) * 4);
    int *patch_lock = reinterpret_cast<int *>(dyn + rows_doubles);
    double *patch = dyn + rows_doubles + 2;               // [64][WT_PITCH] write-out patch (shared; optional)
    if (threadIdx.x == 0) *patch_lock = 0;                // ordered by the barrier below / first use
    // Workgroups go round-robin over the 8 XCDs (one L2 each): give every XCD one contiguous
    // range of the work, so that the 64-individual blocks of a tile -- same weights, same score
    // rows -- meet in one L2.
    if (p.run_if) {
        if (__hip_atomic_load(p.run_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(p.rerun_count, 1);
    }
    const unsigned per_xcd = gridDim.x >> 3;
    const unsigned v = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (v >= p.n_work) return;
    const int tile_idx = (int)(v / (unsigned)p.nquad);
    const int ind0 = ((int)(v % (unsigned)p.nquad) * WLOD_WAVES + wave) * WAVE;
    const bool active = ind0 < p.ind_count;      // the last workgroup of a tile may have idle waves
    const int2 td = p.tiles[tile_idx];
    const ChrDev c = p.chrs[td.x];
    const int s0 = td.y;
    const int64_t col = (int64_t)p.ind_begin + ind0 + lane;
    const uint32_t *gcol = packed + packed_index(0, col, p.nwordrows);
    const int64_t G0 = c.loc_base + GOFF + s0;
    const bool has = lane < TILE && s0 + lane < c.nloci && p.valid[c.loc_base + s0 + lane] != 0;
    const uint32_t vm = (uint32_t)__ballot(has);
    if (!FROM_SCORES && vm != 0) {   // same for every wave of the workgroup
        const double2 *src = reinterpret_cast<const double2 *>(wtab + G0 * 4);
        double2 *dst = reinterpret_cast<double2 *>(rows);
        for (int k = threadIdx.x; k < (W + TILE - 1) * 2; k += WLOD_WAVES * WAVE) dst[k] = src[k];
    }
    __syncthreads();
    if (!active) return;
#pragma unroll 1
    for (int grp = 0; grp < TILE / R; grp++) {
        double acc[R];
        const uint32_t gm = (vm >> (grp * R)) & ((1u << R) - 1u);
        if (gm != 0) {
            if (SMALLW) {          // W < R
                const double *Dp = D + (c.loc_base + s0 + grp * R) * (int64_t)W;
                const int64_t G = G0 + grp * R;
                if (FROM_SCORES) {
                    const double *tp = wtab + ((col >> 6) * p.score_rows + G) * WAVE + (col & 63);
                    wlod_group_small<R>([&](int i) -> double { return tp[i * WAVE]; }, Dp, W, acc);
                } else {
                    const double *rw = rows + grp * R * 4;
                    wlod_group_small<R>([&](int i) -> double {
                        const uint32_t word = gcol[((G + i) >> 4) * WAVE];
                        return rw[i * 4 + ((word >> (2 * (uint32_t)((G + i) & 15))) & 3u)];
                    }, Dp, W, acc);
                }
            } else if (GL_RING) {
                // block-aligned shard (host-checked): the wave's 64 lanes are one block of the matrix
                const int64_t blk = ((int64_t)p.ind_begin + ind0) >> 6;
                const uint32_t ring = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double *)dyn +
                                      WLOD_GL_RING_OFF + (uint32_t)wave * (GARLIC_WLOD_GL_RING_ROWS * WAVE * 8u);
                wlod_group_gl<R>(ring, wtab + (blk * p.score_rows + (G0 + grp * R)) * WAVE, lane,
                                 D + (c.loc_base + s0 + grp * R) * (int64_t)W, W, acc);
            } else if (FROM_SCORES)
                wlod_group_scores<R>(wtab + ((col >> 6) * p.score_rows) * WAVE + (col & 63), G0 + grp * R,
                                     D + (c.loc_base + s0 + grp * R) * (int64_t)W, W, acc);
            else
                wlod_group<R>(rows + grp * R * 4, gcol, G0 + grp * R,
                              D + (c.loc_base + s0 + grp * R) * (int64_t)W, W, acc);
        }
        wlod_write_group<R, ALIGNED16>(acc, gm, c, p, out, patch, patch_lock, ind0, s0, grp, lane, td.x);
    }
}

// The two kernels differ in their register budget (an attribute cannot depend on a template
// argument): 64 VGPRs = 8 waves per SIMD for the hand-scheduled variant, whose scalar weight loads
// need the occupancy; 96 for the term-matrix variant, which spilled at 64.
template <int R, bool ALIGNED16>
__global__ void __launch_bounds__(WLOD_WAVES * WAVE) __attribute__((amdgpu_num_vgpr(64)))
wlod_tile_kernel(const uint32_t *__restrict__ packed, const double *__restrict__ wtab,
                 const double *__restrict__ D, double *__restrict__ out, WlodArgs p)
{
    wlod_tile_body<R, ALIGNED16, false>(packed, wtab, D, out, p);
}

// windows narrower than the group (wlod_group_small): plain scores / term matrix
template <int R, bool ALIGNED16>
__global__ void __launch_bounds__(WLOD_WAVES * WAVE)
wlod_tile_small_kernel(const uint32_t *__restrict__ packed, const double *__restrict__ wtab,
                       const double *__restrict__ D, double *__restrict__ out, WlodArgs p)
{
    wlod_tile_body<R, ALIGNED16, false, false, true>(packed, wtab, D, out, p);
}

template <int R, bool ALIGNED16>
__global__ void __launch_bounds__(WLOD_WAVES * WAVE)
wlod_tile_small_gl_kernel(const uint32_t *__restrict__ packed, const double *__restrict__ terms,
                          const double *__restrict__ D, double *__restrict__ out, WlodArgs p)
{
    wlod_tile_body<R, ALIGNED16, true, false, true>(packed, terms, D, out, p);
}

// Two 64-individual blocks per wave (GARLIC_WLOD2_LOOP_ASM): the 16 weights of a step multiply both blocks'
// scores.  The scalar data path returns one dword per cycle and CU; at 16 weights (32 dwords) per 32
// FP64 operations it, not the FP64 pipe, paced wlod_tile_kernel (0.61 of the FP64 peak; 0.78 with every
// other load left out, measured).  Here a wave spends 64 operations on the same 32 dwords.
template <int R>
__device__ __forceinline__ void wlod_group2(const double *rows, const uint32_t *packed, int64_t colA, int64_t colB,
                                            int64_t nwordrows, int64_t G, const double *Ds, int W, double (&acc)[R],
                                            double (&bcc)[R], bool touch_ahead)
{
    static_assert(R == 16, "the hand-scheduled loop keeps 16 weights per step in SGPRs");
    uint64_t gaddr = reinterpret_cast<uint64_t>(packed + packed_index(G >> 4, colA, nwordrows));
    uint64_t gaddrb = reinterpret_cast<uint64_t>(packed + packed_index(G >> 4, colB, nwordrows));
    uint32_t bit = 2 * (uint32_t)(G & 15);
    double sc, scn, scb, scnb, t0, t1;
    uint32_t vt, vtb, word, nextw, wordb, nextwb, vd;
    uint32_t n = (uint32_t)(W - (R - 1));
    uint32_t row = (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) double *)rows);
    const double *dp = Ds - (R - 1);
    const uint32_t stride = (uint32_t)(W + 1) * 8u;
    asm volatile(GARLIC_WLOD2_LOOP_ASM
                 : [a0] "=&v"(acc[0]), [a1] "=&v"(acc[1]), [a2] "=&v"(acc[2]), [a3] "=&v"(acc[3]),
                   [a4] "=&v"(acc[4]), [a5] "=&v"(acc[5]), [a6] "=&v"(acc[6]), [a7] "=&v"(acc[7]),
                   [a8] "=&v"(acc[8]), [a9] "=&v"(acc[9]), [a10] "=&v"(acc[10]), [a11] "=&v"(acc[11]),
                   [a12] "=&v"(acc[12]), [a13] "=&v"(acc[13]), [a14] "=&v"(acc[14]), [a15] "=&v"(acc[15]),
                   [b0] "=&v"(bcc[0]), [b1] "=&v"(bcc[1]), [b2] "=&v"(bcc[2]), [b3] "=&v"(bcc[3]),
                   [b4] "=&v"(bcc[4]), [b5] "=&v"(bcc[5]), [b6] "=&v"(bcc[6]), [b7] "=&v"(bcc[7]),
                   [b8] "=&v"(bcc[8]), [b9] "=&v"(bcc[9]), [b10] "=&v"(bcc[10]), [b11] "=&v"(bcc[11]),
                   [b12] "=&v"(bcc[12]), [b13] "=&v"(bcc[13]), [b14] "=&v"(bcc[14]), [b15] "=&v"(bcc[15]),
                   [sc] "=&v"(sc), [scn] "=&v"(scn), [scb] "=&v"(scb), [scnb] "=&v"(scnb), [t0] "=&v"(t0), [t1] "=&v"(t1),
                   [vt] "=&v"(vt), [vtb] "=&v"(vtb), [word] "=&v"(word), [nextw] "=&v"(nextw), [wordb] "=&v"(wordb),
                   [nextwb] "=&v"(nextwb), [gaddr] "+v"(gaddr), [gaddrb] "+v"(gaddrb), [bit] "+s"(bit),
                   [row] "+s"(row), [n] "+s"(n), [vd] "=&v"(vd)
                 : [dp] "s"(dp), [stride] "s"(stride), [rowbytes] "s"((uint64_t)(WAVE * 4)), [vz] "v"((uint32_t)GARLIC_WLOD_PFW * stride),
                   [pfon] "s"(__builtin_amdgcn_readfirstlane(touch_ahead && W <= GARLIC_WLOD_PFW_MAX_W ? 1 : 0))
                 : GARLIC_WLOD_LOOP_CLOBBERS);
}

constexpr int WLOD2_BLOCKS = 2 * WLOD_WAVES;   // 64-individual blocks per workgroup of the two-block kernel

template <int R, bool ALIGNED16>
__global__ void __launch_bounds__(WLOD_WAVES * WAVE, 5)   // 5 waves per SIMD: at most 96 VGPRs
wlod_tile2_kernel(const uint32_t *__restrict__ packed, const double *__restrict__ wtab,
                  const double *__restrict__ D, double *__restrict__ out, WlodArgs p)
{   // as wlod_tile_body<R, ALIGNED16, false>, a wave owning the blocks 2w and 2w+1 of its workgroup's eight
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    const int lane = threadIdx.x & (WAVE - 1), W = p.winsize;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *rows = dyn;                                   // [W + TILE][4]
    const size_t rows_doubles = (size_t)((W + TILE) * 4);
    int *patch_lock = reinterpret_cast<int *>(dyn + rows_doubles);
    double *patch = dyn + rows_doubles + 2;
    if (threadIdx.x == 0) *patch_lock = 0;
    const unsigned per_xcd = gridDim.x >> 3;
    const unsigned v = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (v >= p.n_work) return;
    const int tile_idx = (int)(v / (unsigned)p.nquad);
    const int ind0A = ((int)(v % (unsigned)p.nquad) * WLOD2_BLOCKS + 2 * wave) * WAVE, ind0B = ind0A + WAVE;
    const bool activeA = ind0A < p.ind_count, activeB = ind0B < p.ind_count;
    const int2 td = p.tiles[tile_idx];
    const ChrDev c = p.chrs[td.x];
    const int s0 = td.y;
    const int64_t colA = (int64_t)p.ind_begin + ind0A + lane;
    const int64_t colB = activeB ? colA + WAVE : colA;   // no second block: the first one again, results dropped
    const int64_t G0 = c.loc_base + GOFF + s0;
    const bool has = lane < TILE && s0 + lane < c.nloci && p.valid[c.loc_base + s0 + lane] != 0;
    const uint32_t vm = (uint32_t)__ballot(has);
    if (vm != 0) {
        const double2 *src = reinterpret_cast<const double2 *>(wtab + G0 * 4);
        double2 *dst = reinterpret_cast<double2 *>(rows);
        for (int k = threadIdx.x; k < (W + TILE - 1) * 2; k += WLOD_WAVES * WAVE) dst[k] = src[k];
    }
    __syncthreads();
    if (!activeA) return;
#pragma unroll 1
    for (int grp = 0; grp < TILE / R; grp++) {
        double acc[R], bcc[R];
        const uint32_t gm = (vm >> (grp * R)) & ((1u << R) - 1u);
        if (gm != 0)
            wlod_group2<R>(rows + grp * R * 4, packed, colA, colB, p.nwordrows, G0 + grp * R,
                           D + (c.loc_base + s0 + grp * R) * (int64_t)W, W, acc, bcc, (p.use_patch & 2) == 0);
        wlod_write_group<R, ALIGNED16>(acc, gm, c, p, out, patch, patch_lock, ind0A, s0, grp, lane, td.x);
        if (activeB) wlod_write_group<R, ALIGNED16>(bcc, gm, c, p, out, patch, patch_lock, ind0B, s0, grp, lane, td.x);
    }
}

// ... and the hand-scheduled term-matrix variant (per-wave LDS rings): 64 VGPRs again
template <int R, bool ALIGNED16>
__global__ void __launch_bounds__(WLOD_WAVES * WAVE) __attribute__((amdgpu_num_vgpr(64)))
wlod_tile_glring_kernel(const uint32_t *__restrict__ packed, const double *__restrict__ terms,
                        const double *__restrict__ D, double *__restrict__ out, WlodArgs p)
{
    wlod_tile_body<R, ALIGNED16, true, true>(packed, terms, D, out, p);
}

template <int R, bool ALIGNED16>
__global__ void __launch_bounds__(WLOD_WAVES * WAVE) __attribute__((amdgpu_num_vgpr(96)))
wlod_tile_gl_kernel(const uint32_t *__restrict__ packed, const double *__restrict__ terms,
                    const double *__restrict__ D, double *__restrict__ out, WlodArgs p)
{
    wlod_tile_body<R, ALIGNED16, true>(packed, terms, D, out, p);
}

// D[l][j] = 1.0 / LD[l - j][j] for the windows s = l - j of SNP l's own chromosome [lo, hi)
__global__ void skew_reciprocal_kernel(const double *__restrict__ ld, double *__restrict__ D,
                                       int64_t lo, int64_t hi, int W)
{
    const int64_t n = (hi - lo) * W;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t l = lo + i / W;
        const int j = (int)(i % W);
        const int64_t s = l - j;
        D[l * W + j] = (s >= lo) ? reciprocal_x86(ld[s * W + j]) : 0.0;
    }
}

// ... and back: rld[s][j] = D[s + j][j] for the generic wLOD kernel, when it is needed (narrow windows, unaligned
// shards); windows running over the chromosome's end are never scored
__global__ void unskew_kernel(const double *__restrict__ D, double *__restrict__ rld, int64_t lo, int64_t hi, int W)
{
    const int64_t n = (hi - lo) * W;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t s = lo + i / W;
        const int j = (int)(i % W);
        rld[s * W + j] = (s + j < hi) ? D[(s + j) * W + j] : 0.0;
    }
}

// Is any SCORED window exactly -9999.0?  (The tuned chains and the reference's by-value test of
// garlic-roh.cpp:79 differ only from such a window on; everything up to the first one is identical, so the
// tuned chain's own output answers the question.)  One workgroup per (run, 64-individual block) item.
__global__ void __launch_bounds__(256)
sentinel_scan_kernel(const ChainItem *__restrict__ items, const ChrDev *__restrict__ chrs,
                     const double *__restrict__ out, int ind_count, int *__restrict__ flag)
{
    const ChainItem it = items[blockIdx.x];
    if (it.chr < 0) return;
    const ChrDev c = chrs[it.chr];
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x >> 6;
    const int rows = min(WAVE, ind_count - it.ind0);
    bool found = false;
    for (int r = wave; r < rows; r += 4) {
        const double *row = out + c.out_base + (int64_t)(it.ind0 + r) * c.out_pitch;
        for (int s = it.a + lane; s <= it.b; s += WAVE) found |= (row[s] == MISSING_D);
    }
    if (__any(found) && lane == 0) atomicOr(flag, 1);
}

// ---- KDE feed: ordered compaction of every step-th scored window (garlic-data.cpp:2026-2069).
// One wavefront per (chromosome, individual) row; lanes walk the sampled loci 64 at a time and
// rank the keepers with a ballot (integer work: exact whatever the order).
__device__ __forceinline__ bool feed_keep(double x) { return x != MISSING_D && !(x != x); }

// rows: (chromosome, r) pairs, r = 0 .. nrows-1; individual = ind_list ? ind_list[r] : r
// (convertSubsetWinData2DoubleData, garlic-data.cpp:2071-2150: the same loops over randInd[])
__global__ void __launch_bounds__(WAVE)
feed_count_kernel(const double *scores, const ChrDev *chrs, int nchr, int nrows, const int32_t *ind_list, int step,
                  int64_t *row_counts)
{
    const int row = blockIdx.x; // chr * nrows + r
    const ChrDev c = chrs[row / nrows];
    const int ind = ind_list ? ind_list[row % nrows] : row % nrows;
    const double *src = scores + c.out_base + (int64_t)ind * c.out_pitch;
    const int nsamp = (c.nloci + step - 1) / step;
    int cnt = 0;
    for (int k = threadIdx.x; k < nsamp; k += WAVE) cnt += feed_keep(src[(int64_t)k * step]) ? 1 : 0;
    cnt = wave_inclusive_scan(cnt);
    if (threadIdx.x == WAVE - 1) row_counts[row] = cnt;
}

__global__ void __launch_bounds__(WAVE)
feed_write_kernel(const double *scores, const ChrDev *chrs, int nchr, int nrows, const int32_t *ind_list, int step,
                  const int64_t *row_offsets, double *feed)
{
    const int row = blockIdx.x;
    const ChrDev c = chrs[row / nrows];
    const int ind = ind_list ? ind_list[row % nrows] : row % nrows;
    const double *src = scores + c.out_base + (int64_t)ind * c.out_pitch;
    const int nsamp = (c.nloci + step - 1) / step;
    int64_t base = row_offsets[row];
    for (int k0 = 0; k0 < nsamp; k0 += WAVE) {
        const int k = k0 + threadIdx.x;
        const double x = (k < nsamp) ? src[(int64_t)k * step] : MISSING_D;
        const bool keep = (k < nsamp) && feed_keep(x);
        const unsigned long long m = __ballot(keep);
        const int rank = __popcll(m & ((1ull << threadIdx.x) - 1ull));
        if (keep) feed[base + rank] = x;
        base += __popcll(m);
    }
}

// ---- ROH coverage counting: first half of assembleROHWindows (garlic-roh.cpp:446-454).
//   inWin[l] = #{ windows w in (l - W, l] of this individual with score >= cutoff }
// i.e. how many above-cutoff windows cover SNP l (MISSING = -9999 and NaN never qualify; windows
// that would run past the chromosome end cover only the SNPs that exist).  Integer work: per
// (row, 2048-SNP segment) one workgroup marks the windows of the segment and of the W-1 SNPs in
// front of it, takes a prefix count and differences it.  Scores stay on the device; 2 bytes per
// (individual, SNP) come back instead of 8.
constexpr int COV_SEG = 8192, COV_THREADS = 256;   // (2048-window segments: 1.25 M workgroups at 2M x 1280, and the kernel ran at the pace of their dispatch)
__global__ void __launch_bounds__(COV_THREADS)
roh_coverage_kernel(const double *__restrict__ scores, const ChrDev *__restrict__ chrs,
                    const ChrDev *__restrict__ ochrs, const int32_t *__restrict__ seg_base, int nchr,
                    int nind, int W, double cutoff, int16_t *__restrict__ inwin, int vec_ok)
{
    // (16-bit counts: 16.5 KB per workgroup, eight workgroups per CU instead of four -- measured: no faster, 5.3 ms at
    // 2M x 1280 either way; 4.8 TB/s of mixed reads and writes is what the kernel moves)
    extern __shared__ uint16_t pre[];                   // [halo + COV_SEG + 1] inclusive prefix counts
    __shared__ int32_t part[COV_THREADS];
    int chr = 0;
    while (chr + 1 < nchr && (int)blockIdx.x >= seg_base[chr + 1]) chr++;
    const ChrDev c = chrs[chr], oc = ochrs[chr];
    const int seg0 = ((int)blockIdx.x - seg_base[chr]) * COV_SEG;
    const int ind = blockIdx.y;
    const double *row = scores + c.out_base + (int64_t)ind * c.out_pitch;
    const int halo = W - 1;
    const int first = seg0 - halo;                      // window index of pre[1]
    const int n = halo + min(COV_SEG, c.nloci - seg0);  // windows looked at
    // each wave marks a contiguous quarter, 64 consecutive windows per step (coalesced 512-B reads; a thread per
    // contiguous chunk read a line nine times over), counting along with ballots; then the quarters are chained
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x >> 6;
    const int per = ((n + 3) / 4 + WAVE - 1) / WAVE * WAVE;     // windows per wave, whole steps
    const int lo = wave * per, hi = min(n, lo + per);
    int cnt = 0;
    // eight steps' scores requested before the first is looked at (one load in flight per wave ran at the pace of
    // the memory latency: 0.55 of the HBM rate)
    for (int k0 = lo; k0 < hi; k0 += 8 * WAVE) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k = k0 + u * WAVE + lane, w = first + k;
            v[u] = (k < hi && w >= 0) ? __builtin_nontemporal_load(row + w) : MISSING_D;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k = k0 + u * WAVE + lane, w = first + k;
            const bool q = (k < hi) && (w >= 0) && (v[u] >= cutoff);    // NaN >= x is false
            const uint64_t m = __ballot(q);
            if (k < hi) pre[k + 1] = (uint16_t)(cnt + __popcll(m & (((uint64_t)2 << lane) - 1)));   // inclusive count
            cnt += __popcll(m);
        }
    }
    if (lane == 0) part[wave] = cnt;
    if (threadIdx.x == 0) pre[0] = 0;
    __syncthreads();
    int offset = 0;
    for (int t = 0; t < wave; t++) offset += part[t];
    if (offset)
        for (int k = lo + lane; k < hi; k += WAVE) pre[k + 1] = (uint16_t)(pre[k + 1] + offset);
    __syncthreads();
    int16_t *orow = inwin + oc.out_base + (int64_t)ind * oc.out_pitch;
    // SNP l = first + k is covered by windows l-W+1 .. l = positions k-halo .. k
    int done = 0;                                       // SNPs of the segment written eight at a time
    if (vec_ok) {
        // rows 16-B aligned (the caller's inwin_pitch_align a multiple of 8; the segment starts at a multiple of 8192):
        // eight counts per lane and store -- 1 KB per wave instruction instead of 128 B (a vector-memory instruction
        // costs the CU ~50 cycles whatever it moves: the 2-B stores were as many instructions as the loads)
        const int nseg = n - halo;
        done = nseg & ~7;
        uint4 *o4 = reinterpret_cast<uint4 *>(orow + seg0);
        for (int j = 8 * (int)threadIdx.x; j < done; j += 8 * COV_THREADS) {
            const int k = halo + j;
            uint32_t w[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t a = (uint32_t)((int)pre[k + 2 * u + 1] - (int)pre[k + 2 * u - halo]) & 0xffffu;
                const uint32_t b = (uint32_t)((int)pre[k + 2 * u + 2] - (int)pre[k + 2 * u + 1 - halo]) & 0xffffu;
                w[u] = a | (b << 16);
            }
            o4[j >> 3] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    for (int k = halo + done + (int)threadIdx.x; k < n; k += COV_THREADS)
        orow[first + k] = (int16_t)((int)pre[k + 1] - (int)pre[k - halo]);
}

} // namespace garlic
