// libgarlic_hip.so -- C ABI (include/garlic_hip.h) over the gfx950 kernels in lod_kernels.hpp.
//
// Host side of the hot path: what calcLODWindows / calcLOD do around the inner loop
// (src/garlic-roh.cpp:18-44, 279-309) -- per-chromosome bookkeeping, the per-SNP term table
// (lod(), src/garlic-roh.cpp:355-386, evaluated once per SNP and genotype with the HOST libm so
// that log10 is the very function the reference calls), segment -> run -> work-list planning --
// and the launches.  There is no CPU fallback: without a HIP device every compute call fails.
#include "../../include/garlic_hip.h"
#include "lod_kernels.hpp"
#include "variant_kernels.hpp"
#include "ld_kernels.hpp"
#include "tgls_ring_kernel.hpp"
#include "wlod_strip_kernel.hpp"
#include "wlod_small_kernel.hpp"
#include "coverage_kernel.hpp"
#include "roh_segments_kernel.hpp"
#include "feed_kernel.hpp"

#include <algorithm>
#include <chrono>
#include <functional>
#include <mutex>
#include <queue>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

using namespace garlic;

namespace {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(e_ == hipErrorOutOfMemory ? GARLIC_ERR_NOMEM : GARLIC_ERR_HIP,      \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,   \
                        __LINE__);                                                          \
    } while (0)

int score_pool_trim();    // idle score buffers (garlic_device_free keeps them mapped) give their memory back

template <class T> struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    int reserve(size_t n)
    {
        if (n <= cap) return GARLIC_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        if (hipMalloc(reinterpret_cast<void **>(&p), n * sizeof(T)) != hipSuccess) {
            (void)hipGetLastError();
            p = nullptr;
            (void)score_pool_trim();                 // out of memory with score buffers idle in the pool: once more without them
        } else {
            cap = n;
            return GARLIC_OK;
        }
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p), n * sizeof(T)));
        cap = n;
        return GARLIC_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// lod(), src/garlic-roh.cpp:355-386.  Host arithmetic, host libm: identical to the reference.
double host_lod(int genotype, double freq, double error)
{
    double autozygous = 1, nonAutozygous = 1;
    if (freq == 0 || freq == 1) {
    } else if (genotype == 0) {
        nonAutozygous = (1 - freq) * (1 - freq);
        autozygous = (1 - error) * (1 - freq) + error * nonAutozygous;
    } else if (genotype == 1) {
        nonAutozygous = 2 * (freq) * (1 - freq);
        autozygous = error * nonAutozygous;
    } else if (genotype == 2) {
        nonAutozygous = (freq) * (freq);
        autozygous = (1 - error) * (freq) + error * nonAutozygous;
    }
    return log10(autozygous / nonAutozygous);
}

// most negative finite entry of a term table (0 if none is negative)
double min_finite(const double *x, size_t n)
{
    double m = 0.0;
    for (size_t i = 0; i < n; i++)
        if (x[i] < m && x[i] > -1.7976931348623157e308) m = x[i];
    return m;
}

template <class F> void parallel_for(int64_t n, int64_t grain, F f)
{
    unsigned hw = std::thread::hardware_concurrency();
    int nt = (int)std::min<int64_t>(std::min<unsigned>(hw ? hw : 1, 16), (n + grain - 1) / grain);
    if (nt <= 1) { f(0, n); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++)
        th.emplace_back([=] { f(n * t / nt, n * (t + 1) / nt); });
    for (auto &x : th) x.join();
}

} // namespace

struct garlic_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    bool async_device = false;   // garlic_ctx_set_async
    int n_cu = 256;              // compute units of the device (persistent kernels: one or a few workgroups per CU)
    // the dominant kernel of the last HIST calls, one event pair each (asynchronous passes are
    // timed without being waited for one by one): garlic_recent_kernel_ms
    static constexpr int HIST = 32;
    hipEvent_t hist0[HIST] = {}, hist1[HIST] = {};
    int64_t n_calls = 0;
    // glibc's log table on the device (tgls_math.hpp) and the verdict of the start-up comparison of
    // the device's log10 with the host's: 0 not run yet, 1 identical on every probe, -1 differs
    DevBuf<double> d_logtab;
    int log10_state = 0;
};

static int score_alloc(garlic_ctx *ctx, size_t bytes, void **out);   // pooled score memory (below)
static int score_free(garlic_ctx *ctx, void *ptr);

// per-call scratch from the score pool (freed buffers stay mapped there: the next call's request costs no hipMalloc)
template <class T> struct PoolBuf {
    T *p = nullptr;
    garlic_ctx *ctx = nullptr;
    int reserve(garlic_ctx *c, size_t n)
    {
        release();
        void *q = nullptr;
        const int rc = score_alloc(c, std::max<size_t>(n, 1) * sizeof(T), &q);
        if (rc) return rc;
        p = (T *)q;
        ctx = c;
        return GARLIC_OK;
    }
    void release()
    {
        if (p) (void)score_free(ctx, p);
        p = nullptr;
    }
};


struct garlic_panel {
    garlic_ctx *ctx = nullptr;
    int32_t nchr = 0;
    int32_t nind = 0;
    int64_t nloci = 0;
    int64_t nind_pad = 0;
    int64_t nwordrows = 0;
    std::vector<int32_t> chr_nloci;
    std::vector<int64_t> chr_off; // nchr + 1
    // host copies of the small per-SNP inputs
    std::vector<int32_t> pos, cs, ce;
    std::vector<double> gpos, freq;
    bool have_map = false, have_freq = false, have_geno = false, have_gpos = false;
    // device state
    DevBuf<uint32_t> d_packed;
    DevBuf<int32_t> d_pos, d_cs, d_ce;
    DevBuf<int64_t> d_chr_off;
    DevBuf<double> d_tab;
    bool tab_valid = false;
    double tab_error = 0;
    double tab_min = 0, tabgl_min = 0, glterms_min = 0;   // most negative finite term (lod_exact_needed)
    bool tab_all_finite = false;                   // no infinite or NaN term (--error 0, a NaN frequency): every scored window is finite
    // segment boundaries (global loci, ascending), cached per max_gap
    bool seg_valid = false;
    int32_t seg_max_gap = 0;
    std::vector<int64_t> boundaries;
    DevBuf<int32_t> d_blk_counts, d_blk_offsets, d_total;
    DevBuf<int64_t> d_boundaries;
    // per-call scratch
    DevBuf<ChainItem> d_items;
    DevBuf<FeedItem> d_feed_items;                 // thinned feed: (run, FEED_G blocks) items of lod_feed_kernel
    DevBuf<FillItem> d_fill;
    DevBuf<int32_t> d_counter;
    DevBuf<ChrDev> d_chrs;
    DevBuf<int16_t> d_stage16;
    DevBuf<int64_t> d_row_counts;
    // TGLS: dictionary-coded per-genotype error probabilities
    bool have_gl = false;
    std::vector<double> gl_values;                 // code -> error probability
    std::unordered_map<uint64_t, int> gl_code;     // bit pattern -> code
    DevBuf<uint8_t> d_codes;                       // [GOFF+nloci+pad][nind_pad]
    DevBuf<double> d_tabgl;
    bool tabgl_valid = false;
    DevBuf<double> d_glterms;                      // TGLS term matrix [blk][GOFF+nloci+pad][64]
    bool glterms_valid = false, glterms_scaled = false;   // scaled: holds (term * nomut) * norec of (glterms_M, glterms_mu)
    int32_t glterms_M = 0;
    double glterms_mu = 0.0;

    int tabgl_ncodes = 0;
    // TGLS, continuous likelihoods (more distinct values than the dictionary holds): the error
    // probabilities themselves, in the term matrix's layout, and lod() on the device
    bool gl_cont = false;
    DevBuf<double> d_glval;                        // [blk][GOFF+nloci+pad][64]; empty once converted in place
    bool gl_vals_dropped = false;                  // d_glterms owns what was d_glval: terms cannot be rebuilt
    bool gl_cover_required = false;                // after a restart of the upload: every locus must come again
    std::vector<uint8_t> gl_cover;                 // loci uploaded since then
    DevBuf<double> d_freq;                         // [GOFF+nloci+pad], pad rows 0
    bool dfreq_valid = false;
    int gl_terms_by = 0;                           // who built the current terms: 1 device log10, 2 host libm
    // wLOD
    bool have_ld = false, wlod_use_gl = false, rld_valid = false;
    int32_t last_chain_kind = 0;                   // garlic_panel_chain_kind
    int32_t ld_winsize = 0;
    DevBuf<double> d_rld, d_decay, d_stage64;
    DevBuf<uint64_t> d_phase;                      // HapData::firstCopy as bit planes [blk][nloci] (--phased LD)
    uint64_t geno_epoch = 0;                       // bumped by every genotype upload (LD plane cache)
    CovBits cov_pending{nullptr, nullptr, 0.0};    // set by garlic_roh_coverage_fused around a score call: bits, not scores
    bool cov_written = false;                      // ... and a kernel that writes bits took the call
    bool have_phase = false;
    // scratch of the LD-weight kernels, kept between calls (window-size sweeps): at 10M SNPs the six
    // 8-GB allocations and frees of a call cost 9x its kernels.  garlic_panel_release_scratch drops it.
    struct {
        DevBuf<uint64_t> sub, m, h, o;
        DevBuf<int32_t> loc, pair, loc_planes;
        DevBuf<double> hf, fwd, bwd, ld;
        DevBuf<LdSumChr> sum_chrs;
        DevBuf<LdPairChr> pair_chrs;
        // the bit planes (and the per-SNP counts made with them) depend on the genotypes and the LD subsample only, not
        // on the window size: kept across calls (--winsize-multi with --weighted: 3.5 of a call's 32 ms at 10M x 1250)
        uint64_t planes_key = 0;
        bool planes_valid = false;
        // garlic_panel_compute_ld -> garlic_ld_counts: go on to the hr2 table in the pair kernel; -> garlic_ld_finish: it is there
        bool fuse_request = false, fused_done = false;
        void release()
        {
            sub.release(); m.release(); h.release(); o.release(); loc.release(); pair.release(); loc_planes.release();
            hf.release(); fwd.release(); bwd.release(); ld.release(); sum_chrs.release(); pair_chrs.release();
            planes_valid = false;
        }
    } lds;
    // tuned wLOD path: skewed reciprocal weights, per-SNP score rows, window mask, tile index
    DevBuf<double> d_skew, d_wtab;
    DevBuf<uint8_t> d_valid;
    DevBuf<int2> d_tiles, d_segs;        // wLOD work lists: 32-window tiles; WSM_T-window segments (narrow windows)
    DevBuf<WlodStrip> d_strips;
    std::vector<double> h_tab, h_decay;            // host copies the score rows are built from
    bool wtab_valid = false;
    double wtab_error = 0.0, wtab_mu = 0.0;
    int32_t wtab_M = 0;
    int64_t wtab_rows = 0;
    bool decay_valid = false;
    int32_t decay_M = 0;
    double decay_mu = 0;
    // full-score scratch of the host-output and feed calls: pooled score memory, for big unweighted panels chosen by
    // placement at first use (garlic_panel_alloc_scores) -- a caller that hands over host buffers cannot do that itself
    struct ScoreBuf {
        double *p = nullptr;
        size_t cap = 0;
        garlic_ctx *ctx = nullptr;
        int reserve(garlic_ctx *c, size_t n)
        {
            if (n <= cap) return GARLIC_OK;
            release();
            void *q = nullptr;
            int rc = score_alloc(c, n * sizeof(double), &q);
            if (rc) return rc;
            p = (double *)q; cap = n; ctx = c;
            return GARLIC_OK;
        }
        void adopt(garlic_ctx *c, void *q, size_t n) { release(); p = (double *)q; cap = n; ctx = c; }
        void release()
        {
            if (p) (void)score_free(ctx, p);
            p = nullptr; cap = 0;
        }
    } d_out;
    bool placing = false;                          // inside the placement probe of d_out
    DevBuf<double> d_feed;
    // garlic_lod_feed_multi: one set of scratch and one stream per window size of the call, kept for the next call
    struct FeedSlot {
        hipStream_t stream = nullptr;
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        DevBuf<FeedItem> items;
        DevBuf<ChrDev> chrs;
        DevBuf<int32_t> counter;
        DevBuf<int64_t> row_counts;
        DevBuf<double> out, feed;
    };
    std::vector<FeedSlot *> feed_slots;
    garlic_call_stats stats{};
    bool stats_pending = false;                    // event times of the last call not read yet
    int stats_slot = 0;                            // the context's event pair that brackets its dominant kernel
    int64_t n_count_timeouts = 0;                  // garlic_call_stats::n_count_timeouts
    struct Placement { int32_t drawn = 0, rounds = 0; float best_ms = 0, median_ms = 0, worst_ms = 0, target_ms = 0; int32_t reached = 0; };
    Placement placement;                           // the last garlic_panel_alloc_scores on this panel
    // work list of the last call, still on the device: repeated calls with the same arguments
    // (bench steps, window-size sweeps coming back to a size) skip planning and uploads
    struct {
        bool valid = false;
        int mode = -1;
        int32_t W = 0, max_gap = 0, ind_begin = 0, ind_count = 0, pitch_align = 0;
        size_t n_items = 0, n_fill = 0;
        bool wlod_fast = false, wlod_strip = false, feed_kernel = false;
        int32_t thin_step = 0;
        size_t n_feed_items = 0;
        int feed_per_cu = 1;      // persistent workgroups per CU the feed kernel of this plan is launched with (feed_grid)
        uint64_t blocks_hash = 0;                  // 0: every 64-individual block; else a hash of the block subset
        int32_t n_tiles = 0, n_segs = 0, n_strips = 0;
        int64_t n_runs = 0, n_valid = 0;
    } plan;
};

static int ensure_rld(garlic_panel *p);   // plain reciprocals of the LD weights, made when the generic wLOD kernel needs them

// ---- Score buffers.  Where 8 GB of scores sit in VRAM decides between two speeds of lod_chain_kernel at 1M SNPs x
// 1000 individuals (1.36 and 1.62 ms: DESIGN.md section 4, "placement"); a virtual range backed by physical chunks
// of its own (HIP virtual memory management, 1 GB each) was in the fast mode more often than plain hipMalloc
// memory.  Falls back to hipMalloc where the driver has no virtual memory management.
//
// Freed buffers stay MAPPED in a pool and are handed out again for requests they fit: measured on ROCm 7.2 / MI355X,
// a virtual range that is unmapped and given new physical memory loses part of the first kernel's writes
// (tools/exp/alloc_dbg.py, tools/exp/vmm_remap_repro.hip); a buffer that keeps its mapping has nothing to lose, and a
// caller that allocates per sweep reuses the same few ranges instead of growing its address space.  The pool is
// capped (GARLIC_ALLOC_POOL_GB, default a quarter of the device memory); what does not fit is unmapped and its
// physical memory released, its range stays reserved (never mapped again: address space only, reported by
// garlic_device_alloc_stats).
struct ScoreAlloc {
    void *ptr;
    size_t size;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    int device;
    bool pooled;          // free, still mapped
    uint64_t stamp;       // when it was pooled (oldest goes first)
};
static std::mutex g_score_mutex;
static std::vector<ScoreAlloc> g_score_allocs;
static int64_t g_score_retired[16] = {};   // bytes of ranges kept reserved after their memory was released, per device
static uint64_t g_score_clock = 0;

static void release_score_alloc(ScoreAlloc &a, size_t mapped, bool keep_range)
{
    if (mapped) (void)hipMemUnmap(a.ptr, mapped);
    for (auto h : a.handles) (void)hipMemRelease(h);
    if (a.ptr && !keep_range) (void)hipMemAddressFree(a.ptr, a.size);
}

static int score_alloc(garlic_ctx *ctx, size_t bytes, void **out)
{
    *out = nullptr;
    int vmm = 0;
    (void)hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, ctx->device);
    if (vmm && !getenv("GARLIC_ALLOC_PLAIN")) {
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = ctx->device;
        size_t gran = 0;
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) == hipSuccess && gran) {
            const size_t chunk = (((size_t)1 << 30) + gran - 1) / gran * gran;
            const size_t size = (bytes + gran - 1) / gran * gran;
            {   // a pooled buffer that fits without wasting more than an eighth
                std::lock_guard<std::mutex> lock(g_score_mutex);
                int pick = -1;
                for (size_t k = 0; k < g_score_allocs.size(); k++) {
                    const ScoreAlloc &a = g_score_allocs[k];
                    if (a.pooled && a.device == ctx->device && a.size >= size && a.size - size <= size / 8 &&
                        (pick < 0 || a.size < g_score_allocs[(size_t)pick].size))
                        pick = (int)k;
                }
                if (pick >= 0) {
                    g_score_allocs[(size_t)pick].pooled = false;
                    *out = g_score_allocs[(size_t)pick].ptr;
                    return GARLIC_OK;
                }
            }
            ScoreAlloc a{nullptr, size, {}, ctx->device, false, 0};
            size_t mapped = 0;
            bool ok = hipMemAddressReserve(&a.ptr, size, 0, nullptr, 0) == hipSuccess;
            for (size_t off = 0; ok && off < size; off += chunk) {
                const size_t n = std::min(chunk, size - off);
                hipMemGenericAllocationHandle_t h;
                ok = hipMemCreate(&h, n, &prop, 0) == hipSuccess;
                if (!ok) break;
                a.handles.push_back(h);
                ok = hipMemMap((char *)a.ptr + off, n, 0, h, 0) == hipSuccess;
                if (ok) mapped = off + n;
            }
            if (ok) {
                hipMemAccessDesc acc{};
                acc.location = prop.location;
                acc.flags = hipMemAccessFlagsProtReadWrite;
                ok = hipMemSetAccess(a.ptr, size, &acc, 1) == hipSuccess;
            }
            if (ok) {
                *out = a.ptr;
                std::lock_guard<std::mutex> lock(g_score_mutex);
                g_score_allocs.push_back(std::move(a));
                return GARLIC_OK;
            }
            release_score_alloc(a, mapped, mapped != 0);
            if (mapped != 0 && a.ptr) {      // (a partly mapped range stays reserved: garlic_device_alloc_stats counts it)
                std::lock_guard<std::mutex> lock(g_score_mutex);
                g_score_retired[ctx->device < 16 ? ctx->device : 15] += (int64_t)a.size;
            }
            (void)hipGetLastError();
            {   // out of device memory with buffers idle in the pool: give those back and try once more
                bool any = false;
                {
                    std::lock_guard<std::mutex> lock(g_score_mutex);
                    for (const ScoreAlloc &b : g_score_allocs) any = any || (b.pooled && b.device == ctx->device);
                }
                if (any) {
                    (void)score_pool_trim();
                    return score_alloc(ctx, bytes, out);
                }
            }
        }
    }
    HIP_TRY(hipMalloc(out, bytes));
    return GARLIC_OK;
}

namespace {
int score_pool_trim()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return GARLIC_OK;
    (void)hipDeviceSynchronize();
    std::lock_guard<std::mutex> lock(g_score_mutex);
    for (size_t k = 0; k < g_score_allocs.size();) {
        ScoreAlloc &b = g_score_allocs[k];
        if (b.pooled && b.device == dev) {
            release_score_alloc(b, b.size, true);
            g_score_retired[dev < 16 ? dev : 15] += (int64_t)b.size;
            g_score_allocs.erase(g_score_allocs.begin() + (long)k);
        } else k++;
    }
    return GARLIC_OK;
}
}

static int score_free(garlic_ctx *ctx, void *ptr)
{
    {
        // (this device's work first, OUTSIDE the lock: one shard's release must not queue behind another device's kernels)
        bool ours = false;
        {
            std::lock_guard<std::mutex> lock(g_score_mutex);
            for (const ScoreAlloc &a : g_score_allocs) ours = ours || (a.ptr == ptr && !a.pooled);
        }
        if (ours) HIP_TRY(hipDeviceSynchronize());
        size_t total_mem = 0, free_mem = 0;
        if (ours) (void)hipMemGetInfo(&free_mem, &total_mem);
        std::lock_guard<std::mutex> lock(g_score_mutex);
        for (size_t k = 0; k < g_score_allocs.size(); k++)
            if (g_score_allocs[k].ptr == ptr && !g_score_allocs[k].pooled) {
                g_score_allocs[k].pooled = true;
                g_score_allocs[k].stamp = ++g_score_clock;
                // cap the pool: the oldest idle buffers give their memory back (their ranges stay reserved, unmapped for good)
                int64_t cap = (int64_t)(total_mem / 4);
                if (const char *e = getenv("GARLIC_ALLOC_POOL_GB")) cap = (int64_t)(atof(e) * 1073741824.0);
                for (;;) {
                    int64_t pooled = 0;
                    int oldest = -1;
                    for (size_t j = 0; j < g_score_allocs.size(); j++) {
                        const ScoreAlloc &a = g_score_allocs[j];
                        if (!a.pooled || a.device != ctx->device) continue;
                        pooled += (int64_t)a.size;
                        if (oldest < 0 || a.stamp < g_score_allocs[(size_t)oldest].stamp) oldest = (int)j;
                    }
                    if (pooled <= cap || oldest < 0) break;
                    ScoreAlloc &b = g_score_allocs[(size_t)oldest];
                    release_score_alloc(b, b.size, true);
                    g_score_retired[ctx->device < 16 ? ctx->device : 15] += (int64_t)b.size;
                    g_score_allocs.erase(g_score_allocs.begin() + oldest);
                }
                return GARLIC_OK;
            }
    }
    HIP_TRY(hipFree(ptr));
    return GARLIC_OK;
}

namespace {

int set_device(garlic_ctx *ctx)
{
    HIP_TRY(hipSetDevice(ctx->device));
    return GARLIC_OK;
}

// ---- segment boundaries on the device (integer scans), read back once per (map, max_gap)
int ensure_segments(garlic_panel *p, int32_t max_gap)
{
    if (p->seg_valid && p->seg_max_gap == max_gap) return GARLIC_OK;
    garlic_ctx *ctx = p->ctx;
    const int64_t per_block = (int64_t)SEG_BLOCK * SEG_ITEMS;
    const int nblocks = (int)((p->nloci + per_block - 1) / per_block);
    int rc;
    if ((rc = p->d_blk_counts.reserve(nblocks))) return rc;
    if ((rc = p->d_blk_offsets.reserve(nblocks))) return rc;
    if ((rc = p->d_total.reserve(1))) return rc;
    hipLaunchKernelGGL(seg_count_kernel, dim3(nblocks), dim3(SEG_BLOCK), 0, ctx->stream, p->d_pos.p,
                       p->d_chr_off.p, p->d_cs.p, p->d_ce.p, p->nchr, p->nloci, max_gap,
                       p->d_blk_counts.p);
    hipLaunchKernelGGL(seg_scan_kernel, dim3(1), dim3(WAVE), 0, ctx->stream, p->d_blk_counts.p,
                       nblocks, p->d_blk_offsets.p, p->d_total.p);
    int32_t total = 0;
    HIP_TRY(hipMemcpyAsync(&total, p->d_total.p, sizeof total, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (total < p->nchr) return fail(GARLIC_ERR_HIP, "segment scan returned %d boundaries", total);
    if ((rc = p->d_boundaries.reserve((size_t)total))) return rc;
    hipLaunchKernelGGL(seg_compact_kernel, dim3(nblocks), dim3(SEG_BLOCK), 0, ctx->stream,
                       p->d_pos.p, p->d_chr_off.p, p->d_cs.p, p->d_ce.p, p->nchr, p->nloci, max_gap,
                       p->d_blk_offsets.p, p->d_boundaries.p);
    p->boundaries.resize((size_t)total);
    HIP_TRY(hipMemcpyAsync(p->boundaries.data(), p->d_boundaries.p, sizeof(int64_t) * total,
                           hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipGetLastError());
    p->seg_valid = true;
    p->seg_max_gap = max_gap;
    return GARLIC_OK;
}

// ---- per-SNP term table {lod(0), lod(1), lod(2), lod(missing)=+0.0}, host libm
int ensure_term_table(garlic_panel *p, double error)
{
    if (p->tab_valid && memcmp(&p->tab_error, &error, sizeof error) == 0) return GARLIC_OK;
    const int64_t rows = GOFF + p->nloci + GPAD_BACK;
    std::vector<double> tab((size_t)rows * 4, 0.0);
    const double *freq = p->freq.data();
    double *t = tab.data() + (size_t)GOFF * 4;
    parallel_for(p->nloci, 1 << 16, [=](int64_t lo, int64_t hi) {
        for (int64_t l = lo; l < hi; l++) {
            t[l * 4 + 0] = host_lod(0, freq[l], error);
            t[l * 4 + 1] = host_lod(1, freq[l], error);
            t[l * 4 + 2] = host_lod(2, freq[l], error);
            t[l * 4 + 3] = host_lod(-9, freq[l], error);
        }
    });
    int rc;
    if ((rc = p->d_tab.reserve((size_t)rows * 4))) return rc;
    HIP_TRY(hipMemcpyAsync(p->d_tab.p, tab.data(), sizeof(double) * rows * 4, hipMemcpyHostToDevice,
                           p->ctx->stream));
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    p->tab_valid = true;
    p->tab_error = error;
    p->tab_min = min_finite(tab.data(), tab.size());
    p->tab_all_finite = true;
    for (double x : tab)
        if (!std::isfinite(x)) { p->tab_all_finite = false; break; }
    p->h_tab.swap(tab);
    p->wtab_valid = false;
    return GARLIC_OK;
}

struct Layout {
    std::vector<int64_t> base, pitch;
    int64_t total = 0;
};

// thin_step > 0: the layout of the thinned score matrix (KDE feed) -- per chromosome
// ceil(nloci / thin_step) columns instead of nloci
Layout make_layout(const garlic_panel *p, int32_t pitch_align, int32_t nind_out, int32_t thin_step = 0)
{
    Layout L;
    L.base.resize(p->nchr);
    L.pitch.resize(p->nchr);
    int64_t off = 0;
    const int64_t al = std::max(1, pitch_align);
    // pitch_align >= 2: rows are also padded to a multiple of 64 individuals, so every wavefront
    // stores 64 full rows (the pad rows belong to the caller's buffer and are never read back)
    const int64_t rows = (pitch_align >= 2) ? ((int64_t)nind_out + 63) / 64 * 64 : nind_out;
    for (int c = 0; c < p->nchr; c++) {
        const int64_t cols = thin_step > 0 ? ((int64_t)p->chr_nloci[c] + thin_step - 1) / thin_step : p->chr_nloci[c];
        int64_t pitch = (cols + al - 1) / al * al;
        off = (off + al - 1) / al * al;
        L.base[c] = off;
        L.pitch[c] = pitch;
        off += pitch * rows;
    }
    L.total = off;
    return L;
}

struct Run {
    int32_t chr, a, b;
};

// segments -> maximal runs of valid windows for one window size, and the MISSING stretches
void plan_runs(const garlic_panel *p, int32_t W, std::vector<Run> &runs, std::vector<FillItem> &fill,
               int64_t &n_valid)
{
    runs.clear();
    fill.clear();
    n_valid = 0;
    size_t k = 0;
    const size_t nb = p->boundaries.size();
    for (int c = 0; c < p->nchr; c++) {
        const int64_t c0 = p->chr_off[c], c1 = p->chr_off[c + 1];
        int32_t cursor = 0; // first chromosome-local window not yet accounted for
        while (k < nb && p->boundaries[k] < c1) {
            const int64_t s = p->boundaries[k];
            const int64_t e = (k + 1 < nb && p->boundaries[k + 1] < c1) ? p->boundaries[k + 1] : c1;
            k++;
            const int64_t len = e - s;
            if (len >= W) {
                Run r{c, (int32_t)(s - c0), (int32_t)(e - W - c0)};
                if (r.a > cursor) fill.push_back(FillItem{c, cursor, r.a, 0});
                runs.push_back(r);
                n_valid += r.b - r.a + 1;
                cursor = r.b + 1;
            }
        }
        const int32_t n = (int32_t)(c1 - c0);
        if (cursor < n) fill.push_back(FillItem{c, cursor, n, 0});
    }
}

enum Mode { MODE_LOD, MODE_LOD_GL, MODE_WLOD };

// internal (never returned through the ABI): launch_lod was asked for coverage bits (garlic_panel::cov_pending) by a shape
// only the score kernels take; garlic_roh_coverage_fused then computes the scores and counts from them
constexpr int GARLIC_INTERNAL_NO_BITS = -1001;

// Work list of lod_feed_kernel: (run, FEED_G blocks) items, longest runs first (`order`); the runs within reach of
// the longest one run at raised issue priority: their length x one wave's pace is the kernel's critical path.
// blocks: per 64-individual block, 1 = score it (NULL: all nblk of them).
// col0[r]: column of run r's first sampled locus in its chromosome's rows of the sample matrix; < 0: the run holds
// no sampled locus, no item.
void build_feed_items(const std::vector<Run> &runs, const std::vector<int> &order, const std::vector<uint8_t> *blocks,
                      int nblk, const std::vector<int32_t> &col0, std::vector<FeedItem> &items)
{
    items.clear();
    std::vector<int> blk;
    for (int k = 0; k < nblk; k++)
        if (!blocks || (*blocks)[(size_t)k]) blk.push_back(k);
    const int longest = runs.empty() ? 0 : runs[order[0]].b - runs[order[0]].a + 1;
    for (size_t i = 0; i < order.size(); i++) {
        const Run &r = runs[order[i]];
        if (col0[(size_t)order[i]] < 0) continue;
        const int64_t len = r.b - r.a + 1;
        const int prio = (4 * len >= 3 * (int64_t)longest) ? 3 : (2 * len >= longest) ? 2 : (4 * len >= longest) ? 1 : 0;
        for (size_t k = 0; k < blk.size(); k += FEED_G) {
            FeedItem f{r.chr, r.a, r.b, prio, {-1, -1, -1, -1}, col0[(size_t)order[i]], {0, 0, 0}};
            for (size_t w = 0; w < FEED_G && k + w < blk.size(); w++) f.ind0[w] = blk[k + w] * WAVE;
            items.push_back(f);
        }
    }
}

// How many persistent workgroups of lod_feed_kernel per CU.  Not "as many as stay resident": the items are whole runs (a
// chain cannot be cut), a workgroup's pace depends on how many share its CU (measured, 5M x 5k and 2M x 10k: 52 cycles per
// window and wave alone, 80 with one neighbour, 109 with two -- a SIMD gives two chains 1.3 x and three 1.43 x the rate of
// one), and with about as many items as slots the last slots' long items finish alone on an idle chip.  C3 (880 items): three
// per CU 30.8 ms for the four sizes, two per CU 25.6; 2M x 10k (1760 shorter items): three 18.5, two 20.0.  So the launch
// is simulated -- the queue in its order, every CU sharing its pace among the workgroups it holds -- for each count
// that fits, and the shortest one taken.
static double feed_makespan(const std::vector<int32_t> &len, int n_cu, int per_cu)
{
    static const double pace[5] = {0.0, 52.0, 79.6, 109.0, 150.0};       // cycles per window and wave, k workgroups on the CU
    struct Cu { double t; int k; double rem[4]; };
    std::vector<Cu> cus((size_t)n_cu, Cu{0.0, 0, {0, 0, 0, 0}});
    size_t q = 0;
    for (int s = 0; s < per_cu; s++)
        for (int c = 0; c < n_cu && q < len.size(); c++) cus[(size_t)c].rem[cus[(size_t)c].k++] = (double)len[q++];
    typedef std::pair<double, int> Ev;     // (time of the CU's next completion, CU)
    std::priority_queue<Ev, std::vector<Ev>, std::greater<Ev>> heap;
    auto next_of = [&](const Cu &u) {
        double m = u.rem[0];
        for (int i = 1; i < u.k; i++) m = std::min(m, u.rem[i]);
        return u.t + m * pace[u.k];
    };
    for (int c = 0; c < n_cu; c++)
        if (cus[(size_t)c].k) heap.push(Ev(next_of(cus[(size_t)c]), c));
    double end = 0.0;
    while (!heap.empty()) {
        const Ev ev = heap.top();
        heap.pop();
        Cu &u = cus[(size_t)ev.second];
        const double adv = (ev.first - u.t) / pace[u.k];
        u.t = ev.first;
        end = std::max(end, u.t);
        int k = 0;
        for (int i = 0; i < u.k; i++) {
            const double r = u.rem[i] - adv;
            if (r > 0.5) u.rem[k++] = r;
            else if (q < len.size()) u.rem[k++] = (double)len[q++];     // the workgroup pulls the next item
        }
        u.k = k;
        if (k) heap.push(Ev(next_of(u), ev.second));
    }
    return end;
}

int feed_grid(garlic_ctx *ctx, const std::vector<FeedItem> &items, int *grid, int *per_cu_out)
{
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)lod_feed_kernel, FEED_G * WAVE, 0));
    per_cu = std::max(1, std::min(per_cu, 16 / FEED_G));
    const size_t n_items = items.size();
    if (const char *e = getenv("GARLIC_FEED_PER_CU")) per_cu = std::max(1, atoi(e));
    else if (n_items > (size_t)ctx->n_cu && n_items <= (size_t)64 * ctx->n_cu * per_cu) {
        // (fewer items than CUs: one each; very many: whatever order they finish in, the chip stays full)
        std::vector<int32_t> len(n_items);
        for (size_t i = 0; i < n_items; i++) len[i] = items[i].b - items[i].a + 1 + 64;      // (+ an item's fixed costs)
        int best = per_cu;
        double best_t = feed_makespan(len, ctx->n_cu, per_cu);
        for (int k = per_cu - 1; k >= 1; k--) {
            const double t = feed_makespan(len, ctx->n_cu, k);
            if (t < 0.98 * best_t) { best_t = t; best = k; }
        }
        per_cu = best;
    }
    if (per_cu_out) *per_cu_out = per_cu;
    *grid = (int)std::min<size_t>(n_items, (size_t)ctx->n_cu * per_cu);
    return GARLIC_OK;
}

// The reference tests "previous window has no score" by value (garlic-roh.cpp:79); the tuned chains by
// position.  They agree unless a scored window sums to exactly -9999.0, which needs W terms that can add
// up to it: impossible while W * (most negative term) stays above -9999 (a margin covers the rounding of
// the sums).  Otherwise the exact kernel runs (lod_chain_exact_kernel).  Tables / terms must be current.
bool lod_exact_needed(const garlic_panel *p, Mode mode, int32_t W)
{
    if (getenv("GARLIC_EXACT_CHAIN") || getenv("GARLIC_EXACT_CHAIN_ONLY")) return true;
    const double tmin = mode == MODE_LOD ? p->tab_min : (p->gl_cont ? p->glterms_min : p->tabgl_min);
    return (double)W * tmin <= -9990.0;
}

// ---- TGLS: term table per (SNP, error code, genotype), host libm
int ensure_gl_table(garlic_panel *p)
{
    const int ncodes = (int)p->gl_values.size();
    if (ncodes < 1) return fail(GARLIC_ERR_STATE, "use_gl set but no genotype likelihoods were given");
    if (p->tabgl_valid && p->tabgl_ncodes == ncodes) return GARLIC_OK;
    const int64_t rows = GOFF + p->nloci + GPAD_BACK;
    std::vector<double> tab((size_t)rows * ncodes * 4, 0.0);
    const double *freq = p->freq.data();
    const double *val = p->gl_values.data();
    double *t = tab.data() + (size_t)GOFF * ncodes * 4;
    parallel_for(p->nloci, 1 << 12, [=](int64_t lo, int64_t hi) {
        for (int64_t l = lo; l < hi; l++)
            for (int c = 0; c < ncodes; c++) {
                double *e = t + ((size_t)l * ncodes + c) * 4;
                e[0] = host_lod(0, freq[l], val[c]);
                e[1] = host_lod(1, freq[l], val[c]);
                e[2] = host_lod(2, freq[l], val[c]);
                e[3] = host_lod(-9, freq[l], val[c]);
            }
    });
    int rc;
    if ((rc = p->d_tabgl.reserve(tab.size()))) return rc;
    HIP_TRY(hipMemcpyAsync(p->d_tabgl.p, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice,
                           p->ctx->stream));
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    p->tabgl_valid = true;
    p->tabgl_ncodes = ncodes;
    p->tabgl_min = min_finite(tab.data(), tab.size());
    p->glterms_valid = false;
    return GARLIC_OK;
}

// ---- TGLS with continuous likelihoods: glibc's log table on the device, checked against the host
// Probes: both binades __ieee754_log10 hands to log (random mantissas), a dense band around 1 (the
// polynomial branch and its borders), every table cell's ends, random exponents, subnormals and the
// special values.  One kernel, ~2e5 values; the verdict is kept with the context.
int ensure_log10(garlic_ctx *ctx)
{
    if (ctx->log10_state != 0) return GARLIC_OK;
    static const double tab[256] = GLIBC_LOG_TAB;
    int rc;
    if ((rc = ctx->d_logtab.reserve(256))) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->d_logtab.p, tab, sizeof tab, hipMemcpyHostToDevice, ctx->stream));
    std::vector<double> in;
    uint64_t st = 0x9E3779B97F4A7C15ull;
    auto next = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    auto bits = [](uint64_t u) { double d; memcpy(&d, &u, sizeof d); return d; };
    for (double x : {0.0, -0.0, 1.0, -1.0, (double)INFINITY, -(double)INFINITY, (double)NAN, 5e-324, 2.2250738585072014e-308,
                     1.7976931348623157e308, 0.5, 2.0, 10.0, 1e-16})
        in.push_back(x);
    in.push_back(bits(0xFFF8000000000000ull));
    in.push_back(bits(0x7FF0000000000001ull));
    for (uint64_t c : {0x3FEE000000000000ull, 0x3FF1090000000000ull, 0x3FF0000000000000ull, 0x3FE6000000000000ull,
                       0x3FF6000000000000ull, 0x0010000000000000ull})
        for (int d = -16; d <= 16; d++) in.push_back(bits(c + (uint64_t)(int64_t)d));
    for (int cell = 0; cell < 128; cell++)
        for (uint64_t top : {0x3FE0000000000000ull, 0x3FF0000000000000ull})
            for (int d = -2; d <= 2; d++) in.push_back(bits(top + ((uint64_t)cell << 45) + (uint64_t)(int64_t)d));
    for (int k = 0; k < 40000; k++) {
        const uint64_t m = next() & 0x000FFFFFFFFFFFFFull;
        in.push_back(bits(0x3FE0000000000000ull | m));
        in.push_back(bits(0x3FF0000000000000ull | m));
        in.push_back(bits((0x3FF0000000000000ull - (1ull << 49)) + (next() % (3ull << 49))));
        in.push_back(bits(((next() % 2046 + 1) << 52) | m));
        if ((k & 63) == 0) in.push_back(bits(m));
    }
    const int64_t n = (int64_t)in.size();
    DevBuf<double> d_in, d_out;
    auto done = [&](int code) { d_in.release(); d_out.release(); return code; };
    if ((rc = d_in.reserve((size_t)n)) || (rc = d_out.reserve((size_t)n))) return done(rc);
    std::vector<double> out((size_t)n);
    hipError_t e = hipMemcpyAsync(d_in.p, in.data(), sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(log10_probe_kernel, dim3(256), dim3(256), 0, ctx->stream, d_in.p, ctx->d_logtab.p, n, d_out.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out.data(), d_out.p, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "log10 probe: %s", hipGetErrorString(e)));
    int64_t bad = 0;
    for (int64_t i = 0; i < n; i++) {
        const double want = log10(in[(size_t)i]);
        if (memcmp(&want, &out[(size_t)i], sizeof want) != 0) bad++;
    }
    ctx->log10_state = bad == 0 ? 1 : -1;
    if (bad)
        fprintf(stderr, "libgarlic_hip: the host's log10 is not the glibc 2.35 FMA variant the device restates "
                        "(%lld of %lld probes differ); continuous TGLS terms will be computed on the host\n",
                (long long)bad, (long long)n);
    return done(GARLIC_OK);
}

// allele frequencies on the device, in padded row order (pad rows 0 -> term +0.0)
int ensure_dfreq(garlic_panel *p)
{
    if (p->dfreq_valid) return GARLIC_OK;
    const int64_t rows = GOFF + p->nloci + GPAD_BACK;
    std::vector<double> f((size_t)rows, 0.0);
    memcpy(f.data() + GOFF, p->freq.data(), sizeof(double) * (size_t)p->nloci);
    int rc;
    if ((rc = p->d_freq.reserve((size_t)rows))) return rc;
    HIP_TRY(hipMemcpyAsync(p->d_freq.p, f.data(), sizeof(double) * rows, hipMemcpyHostToDevice, p->ctx->stream));
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    p->dfreq_valid = true;
    return GARLIC_OK;
}

// The dictionary is full (or the caller's values are continuous from the start): from here on the
// panel keeps the error probabilities themselves.  What has been coded so far is decoded.
int switch_to_continuous(garlic_panel *p)
{
    if (p->gl_cont) return GARLIC_OK;
    const int64_t rows = GOFF + p->nloci + GPAD_BACK;
    const size_t n = (size_t)rows * p->nind_pad;
    hipStream_t s = p->ctx->stream;
    int rc;
    HIP_TRY(hipStreamSynchronize(s));
    p->d_glterms.release();          // terms of the dictionary the panel leaves behind
    p->glterms_valid = false;
    if ((rc = p->d_glval.reserve(n))) return rc;
    if (p->d_codes.p && !p->gl_values.empty()) {
        std::vector<double> dict(GL_DICT_MAX, 0.0);
        std::copy(p->gl_values.begin(), p->gl_values.end(), dict.begin());
        DevBuf<double> d_dict;
        if ((rc = d_dict.reserve(GL_DICT_MAX))) return rc;
        hipError_t e = hipMemcpyAsync(d_dict.p, dict.data(), sizeof(double) * GL_DICT_MAX, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(gl_decode_kernel, dim3(4096), dim3(256), 0, s, p->d_codes.p, d_dict.p, p->nind_pad, rows,
                               p->d_glval.p);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        d_dict.release();
        if (e != hipSuccess) return fail(GARLIC_ERR_HIP, "set_gl: %s", hipGetErrorString(e));
    } else {
        HIP_TRY(hipMemsetAsync(p->d_glval.p, 0, sizeof(double) * n, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    p->d_codes.release();
    p->d_tabgl.release();
    p->gl_code.clear();
    p->gl_values.clear();
    p->tabgl_valid = false;
    p->gl_cont = true;
    p->gl_vals_dropped = false;
    p->glterms_valid = false;
    return GARLIC_OK;
}

// A new upload after the values were converted in place: the matrix goes back to holding values, all
// zero, and every locus has to come again before the next computation.
int restart_continuous_upload(garlic_panel *p)
{
    if (!p->gl_cont || !p->gl_vals_dropped) return GARLIC_OK;
    const int64_t rows = GOFF + p->nloci + GPAD_BACK;
    const size_t n = (size_t)rows * p->nind_pad;
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    std::swap(p->d_glval.p, p->d_glterms.p);
    std::swap(p->d_glval.cap, p->d_glterms.cap);
    HIP_TRY(hipMemsetAsync(p->d_glval.p, 0, sizeof(double) * n, p->ctx->stream));
    p->gl_vals_dropped = false;
    p->glterms_valid = false;
    p->gl_cover_required = true;
    p->gl_cover.assign((size_t)p->nloci, 0);
    return GARLIC_OK;
}

// terms from values on the host, with the host's own log10: the fallback when the device's restatement
// of glibc's log10 does not reproduce this host's libm (ensure_log10), or GARLIC_TGLS_HOST_TERMS is set
int build_terms_on_host(garlic_panel *p, const double *vals, double *terms)
{
    const int64_t rows = GOFF + p->nloci + GPAD_BACK;
    const int nblk = (int)(p->nind_pad / WAVE);
    const int64_t chunk = std::max<int64_t>(16, (((int64_t)128 << 20) / (8 * WAVE * nblk)) & ~(int64_t)15);
    std::vector<double> hv((size_t)chunk * WAVE * nblk);
    std::vector<uint32_t> hw((size_t)(chunk / 16 + 2) * WAVE * nblk);
    hipStream_t s = p->ctx->stream;
    const double *freq = p->freq.data();
    for (int64_t G0 = GOFF; G0 < GOFF + p->nloci; G0 += chunk) {
        const int64_t G1 = std::min<int64_t>(GOFF + p->nloci, G0 + chunk), nr = G1 - G0;
        const int64_t w0 = G0 >> 4, nw = ((G1 - 1) >> 4) - w0 + 1;
        for (int b = 0; b < nblk; b++) {
            HIP_TRY(hipMemcpyAsync(hv.data() + (size_t)b * chunk * WAVE, vals + ((int64_t)b * rows + G0) * WAVE,
                                   sizeof(double) * nr * WAVE, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(hw.data() + (size_t)b * (chunk / 16 + 2) * WAVE,
                                   p->d_packed.p + ((int64_t)b * p->nwordrows + w0) * WAVE, sizeof(uint32_t) * nw * WAVE,
                                   hipMemcpyDeviceToHost, s));
        }
        HIP_TRY(hipStreamSynchronize(s));
        double *hvp = hv.data();
        const uint32_t *hwp = hw.data();
        parallel_for(nr * nblk, 256, [=](int64_t lo, int64_t hi) {
            for (int64_t k = lo; k < hi; k++) {
                const int64_t b = k / nr, r = k % nr, G = G0 + r;
                double *v = hvp + ((size_t)b * chunk + r) * WAVE;
                const uint32_t *w = hwp + ((size_t)b * (chunk / 16 + 2) + ((G >> 4) - w0)) * WAVE;
                const double f = freq[G - GOFF];
                for (int lane = 0; lane < WAVE; lane++) {
                    const uint32_t code = (w[lane] >> (2 * (int)(G & 15))) & 3u;
                    v[lane] = host_lod(code == 3u ? -9 : (int)code, f, v[lane]);
                }
            }
        });
        for (int b = 0; b < nblk; b++)
            HIP_TRY(hipMemcpyAsync(terms + ((int64_t)b * rows + G0) * WAVE, hv.data() + (size_t)b * chunk * WAVE,
                                   sizeof(double) * nr * WAVE, hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    return GARLIC_OK;
}

// ---- TGLS pass 1: every (SNP, individual) term, once per panel (window-size independent).
// Dictionary-coded likelihoods: returns GARLIC_OK with glterms_valid unset when the matrix does not
// fit -- the caller then keeps the look-up-in-the-chain kernel.  Continuous likelihoods always end
// with a valid matrix (converted in place when a second buffer does not fit) or an error.
// scaled = multiplied in place by the decay factors of (M, mu) for the weighted tile kernel
// (ensure_decay_table first).  Switching between the two forms rebuilds / rescales.
int ensure_gl_terms(garlic_panel *p, bool scaled = false, int32_t M = 0, double mu = 0.0)
{
    const bool same_scale = p->glterms_scaled && p->glterms_M == M && memcmp(&p->glterms_mu, &mu, sizeof mu) == 0;
    if (p->glterms_valid && (scaled ? same_scale : !p->glterms_scaled)) return GARLIC_OK;
    if (!p->gl_cont && getenv("GARLIC_GL_NO_TERMS")) return GARLIC_OK;
    const int64_t rows = GOFF + p->nloci + GPAD_BACK;
    const size_t n = (size_t)rows * p->nind_pad;
    hipStream_t s = p->ctx->stream;
    int rc;
    const bool rebuild = !p->glterms_valid || p->glterms_scaled;   // the raw terms have to be made (again)
    bool fused_scale = false;                                       // ... and were scaled in the same pass
    if (p->gl_cont && rebuild) {
        if (p->gl_vals_dropped)
            return fail(GARLIC_ERR_STATE, "the likelihoods of this panel were converted to terms in place (no room for "
                                          "both); after changing genotypes, frequencies or the weighting they "
                                          "have to be uploaded again (garlic_panel_set_gl over all loci)");
        if (p->gl_cover_required) {
            for (int64_t l = 0; l < p->nloci; l++)
                if (!p->gl_cover[(size_t)l])
                    return fail(GARLIC_ERR_STATE, "likelihood upload restarted: locus %lld has not been uploaded again",
                                (long long)l);
            p->gl_cover_required = false;
        }
        if ((rc = ensure_log10(p->ctx)) || (rc = ensure_dfreq(p))) return rc;
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        if (p->d_glterms.cap < n && (free_b < n * sizeof(double) || free_b - n * sizeof(double) < (size_t)(0.45 * (double)total_b))) {
            HIP_TRY(hipStreamSynchronize(s));      // idle pooled score buffers are not a reason to give the values up
            (void)score_pool_trim();
            HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        }
        // a second matrix only while it leaves plenty of room for the scores (it is what lets the panel
        // switch between raw and weighted terms later); otherwise the values become the terms
        bool separate = p->d_glterms.cap >= n ||
                        (free_b >= n * sizeof(double) && free_b - n * sizeof(double) >= (size_t)(0.45 * (double)total_b));
        if (const char *e = getenv("GARLIC_TGLS_INPLACE")) separate = atoi(e) == 0;
        if (separate && (rc = p->d_glterms.reserve(n))) return rc;
        double *dst = separate ? p->d_glterms.p : p->d_glval.p;
        p->glterms_valid = false;
        const bool on_host = p->ctx->log10_state < 0 || getenv("GARLIC_TGLS_HOST_TERMS");
        DevBuf<unsigned long long> d_minbits;
        if (on_host) {
            if (separate)   // pad rows of the term matrix: +0.0
                HIP_TRY(hipMemsetAsync(dst, 0, sizeof(double) * n, s));
            if ((rc = build_terms_on_host(p, p->d_glval.p, dst))) return rc;
        } else {
            // all rows, pad rows included: their frequency is 0 and their genotypes code 3 -> +0.0; the most negative
            // finite term (lod_exact_needed) is collected along the way
            if ((rc = d_minbits.reserve(GL_MIN_SLOTS))) return rc;
            HIP_TRY(hipMemsetAsync(d_minbits.p, 0, sizeof(unsigned long long) * GL_MIN_SLOTS, s));
            hipLaunchKernelGGL(gl_terms_cont_kernel, dim3((unsigned)((rows + 63) / 64), (unsigned)(p->nind_pad / WAVE)),
                               dim3(256), 0, s, p->d_packed.p, p->nwordrows, p->d_freq.p, p->ctx->d_logtab.p, p->d_glval.p,
                               (int64_t)0, rows, rows, dst, d_minbits.p);
            HIP_TRY(hipGetLastError());
        }
        if (!separate) {   // the term matrix takes the buffer over
            HIP_TRY(hipStreamSynchronize(s));
            p->d_glterms.release();
            std::swap(p->d_glterms.p, p->d_glval.p);
            std::swap(p->d_glterms.cap, p->d_glval.cap);
            p->gl_vals_dropped = true;
        }
        p->gl_terms_by = on_host ? 2 : 1;
        p->glterms_scaled = false;
        if (!on_host) {
            unsigned long long bits[GL_MIN_SLOTS], best = 0;
            hipError_t e = hipMemcpyAsync(bits, d_minbits.p, sizeof bits, hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            d_minbits.release();
            if (e != hipSuccess) return fail(GARLIC_ERR_HIP, "term minimum: %s", hipGetErrorString(e));
            for (unsigned long long b : bits) best = std::max(best, b);
            p->glterms_min = best ? f64_from_bits(best) : 0.0;
        } else {   // most negative finite term, for lod_exact_needed
            constexpr int NB = 1024;
            DevBuf<double> d_part;
            if ((rc = d_part.reserve(NB))) return rc;
            double part[NB];
            hipLaunchKernelGGL(min_finite_kernel, dim3(NB), dim3(256), 0, s, p->d_glterms.p, (int64_t)n, d_part.p);
            hipError_t e = hipMemcpyAsync(part, d_part.p, sizeof part, hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            d_part.release();
            if (e != hipSuccess) return fail(GARLIC_ERR_HIP, "term minimum: %s", hipGetErrorString(e));
            p->glterms_min = min_finite(part, NB);
        }
    } else if (rebuild) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return GARLIC_OK;
        if (p->d_glterms.cap < n && n * sizeof(double) + ((size_t)8 << 30) > free_b) {
            // score buffers idle in the pool (candidates of an earlier placement probe, freed score matrices) are worth
            // nothing next to the term matrix: give them back first.  (Round 4: 70 GB of them left the 10M x 1250 shard
            // without its term matrix -- the chain then looks its terms up, 4.4 x slower, the weighted strip kernel 60 x.)
            HIP_TRY(hipStreamSynchronize(s));
            (void)score_pool_trim();
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return GARLIC_OK;
        }
        if (p->d_glterms.cap < n && n * sizeof(double) + ((size_t)8 << 30) > free_b) {
            // not enough room: the LD scratch the panel keeps for the next window size is worth less
            // than the term matrix (the look-up-in-the-chain kernel is 10x slower)
            HIP_TRY(hipStreamSynchronize(s));
            p->lds.release();
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return GARLIC_OK;
            if (n * sizeof(double) + ((size_t)8 << 30) > free_b) return GARLIC_OK;
        }
        if ((rc = p->d_glterms.reserve(n))) return rc;
        p->glterms_valid = false;
        // pad rows in front of and behind each block's SNPs: 0.0, the term of a missing genotype (the kernel writes the rest)
        for (int64_t b = 0; b < p->nind_pad / WAVE; b++) {
            HIP_TRY(hipMemsetAsync(p->d_glterms.p + (size_t)b * rows * WAVE, 0, sizeof(double) * GOFF * WAVE, s));
            HIP_TRY(hipMemsetAsync(p->d_glterms.p + ((size_t)b * rows + GOFF + p->nloci) * WAVE, 0,
                                   sizeof(double) * (size_t)(rows - GOFF - p->nloci) * WAVE, s));
        }
        VariantArgs a{p->d_packed.p, nullptr, p->d_tabgl.p, p->d_codes.p, nullptr, nullptr, nullptr, nullptr, nullptr,
                      p->nind_pad, p->nwordrows, 0, 0, 0, (int32_t)p->gl_values.size(), 1, nullptr, 0};
        const size_t terms_lds = sizeof(double) * GL_TERMS_S * 4 * (size_t)a.ncodes;      // <= 64 KB (256 codes)
        if (!getenv("GARLIC_GL_TERMS_GATHER")) {
            if (terms_lds > 48 * 1024)
                HIP_TRY(hipFuncSetAttribute((const void *)gl_terms_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)terms_lds));
            // (the weighted kernel's scores in the same pass when that is what is asked for)
            hipLaunchKernelGGL(gl_terms_lds_kernel, dim3((unsigned)((p->nloci + GL_TERMS_S - 1) / GL_TERMS_S)), dim3(256), terms_lds, s,
                               a, p->nloci, rows, (int)(p->nind_pad / WAVE), scaled ? p->d_decay.p : (const double *)nullptr,
                               p->d_glterms.p);
            fused_scale = scaled;
        } else
        hipLaunchKernelGGL(gl_terms_kernel, dim3((unsigned)((p->nloci + 63) / 64), (unsigned)(p->nind_pad / WAVE)),
                           dim3(256), 0, s, a, p->nloci, rows, p->d_glterms.p);
        HIP_TRY(hipGetLastError());
        p->glterms_scaled = false;
        p->gl_terms_by = 0;
    }
    if (scaled && fused_scale) {
        p->glterms_scaled = true;
        p->glterms_M = M;
        p->glterms_mu = mu;
    } else if (scaled) {
        hipLaunchKernelGGL(gl_scale_kernel, dim3(4096), dim3(256), 0, s, p->d_glterms.p, p->d_decay.p, rows,
                           (int64_t)(p->nind_pad / WAVE));
        HIP_TRY(hipGetLastError());
        p->glterms_scaled = true;
        p->glterms_M = M;
        p->glterms_mu = mu;
    }
    p->glterms_valid = true;
    return GARLIC_OK;
}

// ---- wLOD: per-SNP {nomut, norec} (garlic-roh.cpp:134-140, 246-249), host libm exp
int ensure_decay_table(garlic_panel *p, int32_t M, double mu)
{
    if (!p->have_gpos) return fail(GARLIC_ERR_STATE, "wLOD needs genetic positions (set_map with gpos)");
    if (p->decay_valid && p->decay_M == M && memcmp(&p->decay_mu, &mu, sizeof mu) == 0) return GARLIC_OK;
    const int64_t rows = GOFF + p->nloci + GPAD_BACK;
    std::vector<double> dec((size_t)rows * 2, 0.0);
    for (int c = 0; c < p->nchr; c++)
        for (int64_t l = p->chr_off[c]; l < p->chr_off[c + 1]; l++) {
            const bool first = (l == p->chr_off[c]); // locus 0 of a chromosome: absolute position
            const double dP = first ? (double)p->pos[l] : (double)(p->pos[l] - p->pos[l - 1]);
            const double dG = first ? p->gpos[l] : (p->gpos[l] - p->gpos[l - 1]);
            const double Md = M; // the reference passes the int M as a double parameter
            dec[(GOFF + l) * 2 + 0] = exp(-2.0 * Md * mu * dP);
            dec[(GOFF + l) * 2 + 1] = exp(-2.0 * Md * 1 * dG);
        }
    int rc;
    if ((rc = p->d_decay.reserve(dec.size()))) return rc;
    HIP_TRY(hipMemcpyAsync(p->d_decay.p, dec.data(), sizeof(double) * dec.size(), hipMemcpyHostToDevice,
                           p->ctx->stream));
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    p->decay_valid = true;
    p->decay_M = M;
    p->decay_mu = mu;
    p->h_decay.swap(dec);
    p->wtab_valid = false;
    return GARLIC_OK;
}

// ---- wLOD, tuned path: score of every genotype per SNP, (lod * nomut) * norec in the reference's
// order (garlic-roh.cpp:249) -- the same three doubles the per-individual expression multiplies
int ensure_score_rows(garlic_panel *p, double error, int32_t M, double mu, int32_t W)
{
    const int64_t rows = GOFF + p->nloci + std::max<int64_t>(GPAD_BACK, W + 64);
    if (p->wtab_valid && rows <= p->wtab_rows && p->wtab_M == M &&
        memcmp(&p->wtab_error, &error, sizeof error) == 0 && memcmp(&p->wtab_mu, &mu, sizeof mu) == 0)
        return GARLIC_OK;
    std::vector<double> w((size_t)rows * 4, 0.0);
    const double *t = p->h_tab.data(), *d = p->h_decay.data();
    const int64_t have = std::min<int64_t>(rows, GOFF + p->nloci + GPAD_BACK);
    double *wp = w.data();
    parallel_for(have, 1 << 16, [=](int64_t lo, int64_t hi) {
        for (int64_t G = lo; G < hi; G++)
            for (int g = 0; g < 4; g++) wp[G * 4 + g] = (t[G * 4 + g] * d[2 * G]) * d[2 * G + 1];
    });
    int rc;
    if ((rc = p->d_wtab.reserve(w.size()))) return rc;
    HIP_TRY(hipMemcpyAsync(p->d_wtab.p, w.data(), sizeof(double) * w.size(), hipMemcpyHostToDevice,
                           p->ctx->stream));
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    p->wtab_valid = true;
    p->wtab_error = error; p->wtab_M = M; p->wtab_mu = mu; p->wtab_rows = rows;
    return GARLIC_OK;
}

// thin_step > 0 (unweighted scores, device output, pitch_align 32 only): `out` is the thinned matrix
// of make_layout(p, 32, ind_count, thin_step) -- the chain kernel stores only the windows at loci
// 0, thin_step, 2 * thin_step, .. of each chromosome, everything else of that matrix is MISSING.
int launch_lod(garlic_panel *p, Mode mode, int32_t W, double error, int32_t max_gap, int32_t M, double mu,
               int32_t ind_begin, int32_t ind_count, int32_t pitch_align, double *out, int32_t where,
               int32_t thin_step = 0, const std::vector<uint8_t> *blocks = nullptr)
{   // blocks (chain kernels only): per 64-individual block of the call, 1 = score it; rows of the other
    // blocks are left unwritten (the subset feed never reads them)
    garlic_ctx *ctx = p->ctx;
    int rc;
    if ((rc = set_device(ctx))) return rc;
    if (W <= 1) return fail(GARLIC_ERR_INVALID, "SNP window size must be > 1 (got %d)", W);
    if (!p->have_map || !p->have_freq || !p->have_geno)
        return fail(GARLIC_ERR_STATE, "panel needs map, freq and genotypes before computing LOD");
    if (ind_begin < 0 || ind_count < 1 || (int64_t)ind_begin + ind_count > p->nind)
        return fail(GARLIC_ERR_INVALID, "individual range [%d,+%d) outside panel of %d", ind_begin,
                    ind_count, p->nind);
    if (!out) return fail(GARLIC_ERR_INVALID, "out is NULL");
    if (pitch_align < 1) return fail(GARLIC_ERR_INVALID, "pitch_align must be >= 1");
    const bool use_gl = (mode == MODE_LOD_GL) || (mode == MODE_WLOD && p->wlod_use_gl);
    if (thin_step > 0 && (mode != MODE_LOD || where != GARLIC_DEVICE || pitch_align != 32))
        return fail(GARLIC_ERR_INVALID, "internal: thinned output is for unweighted device scores");

    if ((rc = ensure_segments(p, max_gap))) return rc;
    if (use_gl) {
        if (!p->have_gl) return fail(GARLIC_ERR_STATE, "use_gl set but no genotype likelihoods were given");
        if (!p->gl_cont && (rc = ensure_gl_table(p))) return rc;
        if (mode == MODE_LOD_GL && (rc = ensure_gl_terms(p))) return rc;
    } else if ((rc = ensure_term_table(p, error))) return rc;
    if (mode == MODE_WLOD) {
        if (!p->have_ld || p->ld_winsize != W)
            return fail(GARLIC_ERR_STATE, "wLOD needs LD weights for winsize %d (garlic_panel_set_ld)", W);
        if ((rc = ensure_decay_table(p, M, mu))) return rc;
    }
    // tuned wLOD kernels: 16 window accumulators per lane; scores from one LDS row per SNP (plain
    // --error) or from the TGLS score matrix (use_gl); very narrow / very wide windows keep the
    // generic kernel
    // A window sum of exactly -9999.0 is possible (lod_exact_needed): the tuned chain runs first, its scored windows
    // are scanned for that value, and only if one is there the chain that follows the reference to the letter
    // (11-17 x slower) runs instead.  GARLIC_EXACT_CHAIN_ONLY: that chain straight away.
    const bool exact_possible = mode != MODE_WLOD && lod_exact_needed(p, mode, W);
    bool exact = exact_possible && getenv("GARLIC_EXACT_CHAIN_ONLY") != nullptr;
    if (exact_possible && thin_step > 0) return fail(GARLIC_ERR_INVALID, "internal: thinned output with the exact chain");
    // coverage bits instead of scores (garlic_roh_coverage_fused): only the kernels that know how; nothing else may touch
    // `out` (it is not a score buffer then)
    const bool cov_bits = p->cov_pending.bits != nullptr;
    if (cov_bits && mode == MODE_LOD)
        return fail(GARLIC_ERR_STATE, "internal: coverage bits of the unweighted --error scores come from lod_bits_kernel");
    if (cov_bits && mode == MODE_LOD_GL &&
        (exact_possible || where != GARLIC_DEVICE || (ind_begin & (WAVE - 1)) != 0 || getenv("GARLIC_TGLS_NO_RING")))
        return GARLIC_INTERNAL_NO_BITS;      // the TGLS ring chain / the tuned wLOD kernels do not take this shape
    const bool wlod_shape_ok = mode == MODE_WLOD && W + 64 <= GPAD_BACK && !getenv("GARLIC_WLOD_GENERIC") &&
                               (W >= WLOD_R || !getenv("GARLIC_WLOD_SMALL_GENERIC"));
    const bool wlod_small = W < WLOD_R;      // narrower than a window group: wlod_group_small (compiler-scheduled)
    if (wlod_shape_ok && use_gl && (rc = ensure_gl_terms(p, true, M, mu))) return rc;
    const bool wlod_gl = wlod_shape_ok && use_gl && p->glterms_valid && p->glterms_scaled;   // scores from the term matrix
    bool wlod_fast = (wlod_shape_ok && !use_gl) || wlod_gl;                 // tile kernel, either variant
    if (wlod_fast && !wlod_gl && sizeof(double) * (size_t)(W + TILE) * 4 + 16 > 150 * 1024) wlod_fast = false;
    if (wlod_fast && !wlod_gl && (rc = ensure_score_rows(p, error, M, mu, W))) return rc;
    if (mode == MODE_WLOD && !wlod_fast && (rc = ensure_rld(p))) return rc;
    // narrow windows, plain scores: the streaming kernel (wlod_small_kernel.hpp) reads the plain reciprocals, a window's
    // W weights contiguous
    const bool wlod_stream = wlod_fast && wlod_small && !p->cov_pending.bits && !getenv("GARLIC_WLOD_SMALL_TILES");
    if (p->cov_pending.bits && mode == MODE_WLOD && !wlod_fast) return GARLIC_INTERNAL_NO_BITS;
    if (wlod_stream && (rc = ensure_rld(p))) return rc;
    // continuous likelihoods have no code table: the generic kernel takes its terms from the raw matrix
    if (mode == MODE_WLOD && use_gl && p->gl_cont && !wlod_fast && (rc = ensure_gl_terms(p))) return rc;
    // transposed write-out patch only while rows + patch keep 8 workgroups (32 waves) on a CU
    const size_t wlod_rows = wlod_gl ? 0 : sizeof(double) * (size_t)(W + TILE) * 4;
    const size_t wlod_patch = sizeof(double) * (size_t)WAVE * WT_PITCH;
    const bool wlod_use_patch = wlod_rows + 16 + wlod_patch <= 160 * 1024 / 8 && !getenv("GARLIC_WLOD_NO_PATCH");
    // term-matrix variant: the hand-scheduled loop stages the block's term rows through one LDS ring
    // per wave; it needs a block-aligned shard (a wave's 64 lanes = one block of the matrix)
    const bool wlod_gl_ring = wlod_gl && !wlod_small && (ind_begin & (WAVE - 1)) == 0 && !getenv("GARLIC_WLOD_GL_NO_RING");
    const bool ring_patch = !getenv("GARLIC_WLOD_GL_NO_PATCH");
    // ... and with windows narrow enough for WS_WAVES (W <= 113) or WS_WAVES_WIDE (W <= 241) compute waves per workgroup the strip form: the
    // blocks' term rows enter a CU once per strip (wlod_strip_kernel.hpp)
    const int strip_waves = (W + 15 - 16 * WS_WAVES <= 16 || getenv("GARLIC_WLOD_STRIP_NARROW_ONLY")) ? WS_WAVES : WS_WAVES_WIDE;
    const bool wlod_gl_strip = wlod_gl_ring && W + 15 - 16 * strip_waves <= 16 && !getenv("GARLIC_WLOD_GL_NO_STRIP");
    const bool strip_now = wlod_gl_strip;
    const size_t wlod_lds = wlod_gl_ring ? WLOD_GL_RING_OFF + (size_t)WLOD_WAVES * GARLIC_WLOD_GL_RING_ROWS * WAVE * 8
                                   : wlod_rows + 16 + (wlod_use_patch ? wlod_patch : 0);   // 16: the patch lock

    // Host output: the device always computes into the padded layout the tuned kernels need; the
    // rows are copied out into the caller's (possibly dense) layout by strided D2H copies.
    const Layout Lhost = make_layout(p, pitch_align, ind_count);
    const int32_t pitch_align_host = pitch_align;
    if (where == GARLIC_HOST) pitch_align = std::max(pitch_align, 32);
    Layout L = make_layout(p, pitch_align, ind_count, thin_step);
    for (int c = 0; c < p->nchr; c++)
        if (3 * L.pitch[c] * 8 + 512 >= (int64_t)1 << 32)
            return fail(GARLIC_ERR_INVALID, "chromosome %d too long for 32-bit row offsets", c);

    // Thinned output: every wave a chain of its own (feed_kernel.hpp)
    const bool feed_kernel = thin_step > 0;
    uint64_t blocks_hash = 0;
    if (blocks) {
        blocks_hash = 0xCBF29CE484222325ull;
        for (uint8_t b : *blocks) blocks_hash = (blocks_hash ^ (b ? 1u : 2u)) * 0x100000001B3ull;
        blocks_hash |= 1;
    }
    const bool reuse = p->plan.valid && p->plan.blocks_hash == blocks_hash && p->plan.mode == (int)mode && p->plan.W == W &&
                       p->plan.max_gap == max_gap && p->plan.ind_begin == ind_begin &&
                       p->plan.ind_count == ind_count && p->plan.pitch_align == pitch_align &&
                       p->plan.wlod_fast == wlod_fast && p->plan.thin_step == thin_step && p->plan.wlod_strip == wlod_gl_strip &&
                       p->plan.feed_kernel == feed_kernel;
    const int nblk = (ind_count + WAVE - 1) / WAVE;
    std::vector<Run> runs;
    std::vector<FillItem> fill;
    std::vector<ChainItem> items;
    std::vector<FeedItem> feed_items;
    std::vector<ChrDev> chrs;
    int64_t n_valid = p->plan.n_valid;
    size_t n_items = p->plan.n_items, n_fill = p->plan.n_fill, n_feed_items = p->plan.n_feed_items;
    int64_t n_runs = p->plan.n_runs;
    if (!reuse) {
        plan_runs(p, W, runs, fill, n_valid);
        // Work list: (run, 64-individual block) items, longest runs first (LPT); the persistent
        // workgroups of lod_chain_kernel pull them from a device counter.
        std::vector<int> order(runs.size());
        for (size_t i = 0; i < runs.size(); i++) order[i] = (int)i;
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
            return (runs[x].b - runs[x].a) > (runs[y].b - runs[y].a);
        });
        items.reserve(runs.size() * nblk);
        for (size_t i = 0; i < order.size(); i++) {
            const Run &r = runs[order[i]];
            for (int k = 0; k < nblk; k++)
                if (!blocks || (*blocks)[(size_t)k]) items.push_back(ChainItem{r.chr, r.a, r.b, k * WAVE});
        }
        if (feed_kernel) {
            // the thinned score matrix: row = individual, column = locus / step
            std::vector<int32_t> col0(runs.size());
            for (size_t i = 0; i < runs.size(); i++) {
                const int32_t s = (runs[i].a + thin_step - 1) / thin_step;
                col0[i] = (int64_t)s * thin_step <= runs[i].b ? s : -1;
            }
            build_feed_items(runs, order, blocks, nblk, col0, feed_items);
            n_feed_items = feed_items.size();
            if ((rc = p->d_feed_items.reserve(std::max<size_t>(n_feed_items, 1)))) return rc;
        }
        chrs.resize(p->nchr);
        for (int c = 0; c < p->nchr; c++)
            chrs[c] = ChrDev{p->chr_off[c], L.base[c], L.pitch[c], p->chr_nloci[c],
                             (pitch_align >= 2 && 64 * L.pitch[c] * 8 + 512 < ((int64_t)1 << 32)) ? 1 : 0};
        n_items = items.size();
        n_fill = fill.size();
        n_runs = (int64_t)runs.size();
        if ((rc = p->d_chrs.reserve(chrs.size()))) return rc;
        if ((rc = p->d_items.reserve(std::max<size_t>(n_items, 1)))) return rc;
        if ((rc = p->d_fill.reserve(std::max<size_t>(n_fill, 1)))) return rc;
        // [0], [1]: the chain kernel's queue; [2]: sentinel_scan_kernel's flag; [3]: the strip kernel's stall flag;
        // [4]: strip launches repaired by the tile form since the panel was made (garlic_call_stats::n_stall_reruns)
        if (!p->d_counter.p) {
            if ((rc = p->d_counter.reserve(8))) return rc;
            HIP_TRY(hipMemsetAsync(p->d_counter.p, 0, 8 * sizeof(int32_t), ctx->stream));
        }
        p->plan.valid = false;
    }
    std::vector<uint8_t> valid;
    std::vector<int2> tiles, segs;
    if (wlod_fast && !reuse) {
        valid.assign((size_t)p->nloci, 0);
        for (const Run &r : runs)
            memset(valid.data() + p->chr_off[r.chr] + r.a, 1, (size_t)(r.b - r.a + 1));
        for (int c = 0; c < p->nchr; c++)
            for (int s0 = 0; s0 < p->chr_nloci[c]; s0 += TILE) tiles.push_back(make_int2(c, s0));
        p->plan.n_tiles = (int32_t)tiles.size();
        for (int c = 0; c < p->nchr; c++)
            for (int s0 = 0; s0 < p->chr_nloci[c]; s0 += WSM_T) segs.push_back(make_int2(c, s0));
        p->plan.n_segs = (int32_t)segs.size();
        if ((rc = p->d_valid.reserve(valid.size()))) return rc;
        if ((rc = p->d_tiles.reserve(tiles.size())) || (rc = p->d_segs.reserve(segs.size()))) return rc;
    }
    std::vector<WlodStrip> strips;
    if (wlod_gl_strip && !reuse) {
        // strips of 16-window groups: long enough that filling and draining the workgroup's pipeline (~ 8 groups)
        // stays a few percent, short enough for a few thousand work items
        int64_t total_groups = 0;
        for (int c = 0; c < p->nchr; c++) total_groups += (p->chr_nloci[c] + WLOD_R - 1) / WLOD_R;
        const int64_t pairs = (nblk + 1) / 2;
        int64_t per = total_groups * pairs / 8192;
        per = std::min<int64_t>(256, std::max<int64_t>(64, per)) & ~(int64_t)1;
        if (const char *e = getenv("GARLIC_WLOD_STRIP_GROUPS")) per = std::max<int64_t>(1, atol(e));   // tests: many short strips
        for (int c = 0; c < p->nchr; c++) {
            const int ng = (p->chr_nloci[c] + WLOD_R - 1) / WLOD_R;
            for (int g0 = 0; g0 < ng; g0 += (int)per)
                strips.push_back(WlodStrip{c, g0 * WLOD_R, std::min<int>((int)per, ng - g0), 0});
        }
        p->plan.n_strips = (int32_t)strips.size();
        if ((rc = p->d_strips.reserve(strips.size()))) return rc;
    }
    // Persistent workgroups (4 waves each: CHAIN, POST, PRE, COMB), one per CU; items are pulled longest
    // first, so the short runs pack behind the long ones instead of competing with them for HBM
    // bandwidth.
    int workers = ctx->n_cu;
    if (const char *e = getenv("GARLIC_WORKERS")) workers = std::max(1, atoi(e));
    workers = std::min<int>(workers, (int)n_items);

    double *d_out = out;
    if (where == GARLIC_HOST) {
        // big unweighted score scratch: several candidates, the real kernel timed into each, the fastest kept
        // (the same kernel runs 1.36 or 1.62 ms at 1M x 1000 depending on where its scores sit: DESIGN.md section 4)
        if (p->d_out.cap < (size_t)L.total && mode == MODE_LOD && thin_step == 0 && !p->placing &&
            (size_t)L.total * sizeof(double) >= ((size_t)1 << 30) && !getenv("GARLIC_NO_PLACEMENT")) {
            p->placing = true;
            void *best = nullptr;
            rc = garlic_panel_alloc_scores(p, pitch_align, ind_count, W, error, max_gap, 0, &best, nullptr);
            p->placing = false;
            if (rc) return rc;
            p->d_out.adopt(ctx, best, (size_t)L.total);
            return launch_lod(p, mode, W, error, max_gap, M, mu, ind_begin, ind_count, pitch_align_host, out, where, thin_step, blocks);
        }
        if ((rc = p->d_out.reserve(ctx, (size_t)L.total))) return rc;
        d_out = p->d_out.p;
    }
    const bool aligned16 = (pitch_align % 2 == 0) && ((reinterpret_cast<uintptr_t>(d_out) & 15) == 0);

    // Everything this call puts on the stream.  (Replaying a repeated asynchronous pass as one HIP
    // graph -- counter reset, MISSING fill, chain kernel, events -- was measured: no difference, the
    // 1.6 ms kernel hides the launch gaps of the small operations once passes are enqueued back to back.)
    auto enqueue = [&]() -> int {
    HIP_TRY(hipEventRecord(ctx->ev_begin, ctx->stream));
    if (!reuse) {
        HIP_TRY(hipMemcpyAsync(p->d_chrs.p, chrs.data(), sizeof(ChrDev) * chrs.size(),
                               hipMemcpyHostToDevice, ctx->stream));
        if (n_items)
            HIP_TRY(hipMemcpyAsync(p->d_items.p, items.data(), sizeof(ChainItem) * n_items,
                                   hipMemcpyHostToDevice, ctx->stream));
        if (feed_kernel && n_feed_items)
            HIP_TRY(hipMemcpyAsync(p->d_feed_items.p, feed_items.data(), sizeof(FeedItem) * n_feed_items,
                                   hipMemcpyHostToDevice, ctx->stream));
        if (n_fill)
            HIP_TRY(hipMemcpyAsync(p->d_fill.p, fill.data(), sizeof(FillItem) * n_fill,
                                   hipMemcpyHostToDevice, ctx->stream));
        if (wlod_fast) {
            HIP_TRY(hipMemcpyAsync(p->d_valid.p, valid.data(), valid.size(), hipMemcpyHostToDevice,
                                   ctx->stream));
            HIP_TRY(hipMemcpyAsync(p->d_tiles.p, tiles.data(), sizeof(int2) * tiles.size(),
                                   hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(hipMemcpyAsync(p->d_segs.p, segs.data(), sizeof(int2) * segs.size(),
                                   hipMemcpyHostToDevice, ctx->stream));
            if (!strips.empty())
                HIP_TRY(hipMemcpyAsync(p->d_strips.p, strips.data(), sizeof(WlodStrip) * strips.size(),
                                       hipMemcpyHostToDevice, ctx->stream));
        }
    }
    if (thin_step > 0) {          // small matrix: MISSING everywhere, the chain kernel overwrites the scored samples
        hipLaunchKernelGGL(fill_value_kernel, dim3(1024), dim3(256), 0, ctx->stream, d_out, L.total, MISSING_D);
    } else if (n_fill && !wlod_fast && !cov_bits) {   // the tuned wLOD kernel writes MISSING itself
        dim3 grid((unsigned)n_fill, (unsigned)((ind_count + FILL_ROWS - 1) / FILL_ROWS));
        hipLaunchKernelGGL(fill_missing_kernel, grid, dim3(256), 0, ctx->stream, p->d_fill.p,
                           p->d_chrs.p, ind_count, d_out);
    }
    // queue head and exit count of the persistent chain kernel: its last workgroup leaves both at
    // zero, so only a new plan (or a first call) clears them
    if (n_items && (!reuse || mode != MODE_LOD)) HIP_TRY(hipMemsetAsync(p->d_counter.p, 0, 2 * sizeof(int32_t), ctx->stream));
    p->stats_slot = (int)(ctx->n_calls % garlic_ctx::HIST);
    HIP_TRY(hipEventRecord(ctx->hist0[p->stats_slot], ctx->stream));
    if (wlod_fast) {
        // plain --error scores: two blocks per wave (every scalar-loaded weight used twice); the per-genotype
        // variants keep one block per wave (their term rows, not the weights, set their pace)
        const bool two_blocks = !wlod_gl && !wlod_small && !getenv("GARLIC_WLOD_ONE_BLOCK");
        const int per_wg = two_blocks ? WLOD2_BLOCKS : WLOD_WAVES;
        const int nquad = (nblk + per_wg - 1) / per_wg;
        WlodArgs a{p->d_valid.p, p->d_chrs.p, p->d_tiles.p, p->nwordrows, p->nchr, ind_begin, ind_count, W, nquad,
                   (uint32_t)((int64_t)p->plan.n_tiles * nquad), ((wlod_gl_ring ? ring_patch : wlod_use_patch) ? 1 : 0) | (getenv("GARLIC_WLOD_NO_PF") ? 2 : 0),
                   (int64_t)(GOFF + p->nloci + GPAD_BACK), wlod_gl_ring ? 1 : 0, p->cov_pending, nullptr, nullptr};
        const uint32_t *a_packed = p->d_packed.p;
        const double *a_wtab = wlod_gl ? p->d_glterms.p : p->d_wtab.p, *a_skew = p->d_skew.p + SKEW_FRONT;
        const unsigned wl_grid = (a.n_work + 7u) / 8u * 8u;
        const dim3 wl_block(WLOD_WAVES * WAVE);
        if (wlod_lds > 48 * 1024 && two_blocks) {
            const void *fn = aligned16 ? (const void *)wlod_tile2_kernel<WLOD_R, true> : (const void *)wlod_tile2_kernel<WLOD_R, false>;
            HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlod_lds));
        } else if (wlod_lds > 48 * 1024) {
            const void *fn = wlod_gl ? (aligned16 ? (const void *)wlod_tile_gl_kernel<WLOD_R, true>
                                                  : (const void *)wlod_tile_gl_kernel<WLOD_R, false>)
                                     : (aligned16 ? (const void *)wlod_tile_kernel<WLOD_R, true>
                                                  : (const void *)wlod_tile_kernel<WLOD_R, false>);
            HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlod_lds));
        }
        if (wlod_stream && aligned16) {
            // segments of WSM_T windows x eight blocks per workgroup, everything the window loop reads staged in LDS
            const int nquad8 = (nblk + WLOD2_BLOCKS - 1) / WLOD2_BLOCKS;
            WlodArgs as = a;
            as.tiles = p->d_segs.p;
            as.nquad = nquad8;
            as.n_work = (uint32_t)((int64_t)p->plan.n_segs * nquad8);
            as.use_patch = 1;
            const void *fn = wlod_gl ? wlod_stream_small_gl_fn(W) : wlod_stream_small_fn(W);
            const size_t lds = wlod_small_lds_bytes(W);
            if (lds > 48 * 1024) HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const double *a_rld = p->d_rld.p;
            void *kargs[] = {(void *)&a_packed, (void *)&a_wtab, (void *)&a_rld, (void *)&d_out, (void *)&as};
            HIP_TRY(hipLaunchKernel(fn, dim3((as.n_work + 7u) / 8u * 8u), wl_block, kargs, lds, ctx->stream));
        } else if (wlod_small) {
            const void *fn = wlod_gl ? (aligned16 ? (const void *)wlod_tile_small_gl_kernel<WLOD_R, true>
                                                  : (const void *)wlod_tile_small_gl_kernel<WLOD_R, false>)
                                     : (aligned16 ? (const void *)wlod_tile_small_kernel<WLOD_R, true>
                                                  : (const void *)wlod_tile_small_kernel<WLOD_R, false>);
            void *kargs[] = {(void *)&a_packed, (void *)&a_wtab, (void *)&a_skew, (void *)&d_out, (void *)&a};
            HIP_TRY(hipLaunchKernel(fn, dim3(wl_grid), wl_block, kargs, wlod_lds, ctx->stream));
        } else if (strip_now) {
            const int n_pairs = (nblk + 1) / 2;
            WlodStripArgs sa{p->d_valid.p, p->d_chrs.p, p->d_strips.p, p->d_glterms.p, a_skew, d_out,
                             (int64_t)(GOFF + p->nloci + GPAD_BACK), ind_begin, ind_count, W, strip_waves, n_pairs,
                             ring_patch ? 1 : 0, (uint32_t)((int64_t)p->plan.n_strips * n_pairs), p->d_counter.p + 3, p->cov_pending};
            HIP_TRY(hipMemsetAsync(p->d_counter.p + 3, 0, sizeof(int32_t), ctx->stream));
            const unsigned grid = (sa.n_work + 7u) / 8u * 8u;
            const bool wide = strip_waves == WS_WAVES_WIDE;
            // scores into 16-B aligned rows at W <= 113: the 80-VGPR form, three workgroups per CU (wlod_strip_kernel.hpp)
            bool three = !wide && aligned16 && !sa.cov.bits && !getenv("GARLIC_WLOD_STRIP_TWO_PER_CU");
            for (int k = 0; three && k < p->nchr; k++) three = L.pitch[k] * 8 < ((int64_t)1 << 32);
            const void *fn = three ? (const void *)wlod_strip_gl3_kernel
                             : wide ? (aligned16 ? (const void *)wlod_strip_gl_kernel<true, WS_WAVES_WIDE>
                                                 : (const void *)wlod_strip_gl_kernel<false, WS_WAVES_WIDE>)
                                    : (aligned16 ? (const void *)wlod_strip_gl_kernel<true, WS_WAVES>
                                                 : (const void *)wlod_strip_gl_kernel<false, WS_WAVES>);
            const uint32_t strip_lds = three ? WF_LDS_BYTES : WS_LDS_BYTES;
            HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)strip_lds));
            void *kargs[] = {(void *)&sa};
            HIP_TRY(hipLaunchKernel(fn, dim3(grid), dim3((strip_waves + 1) * WAVE), kargs, strip_lds, ctx->stream));
            // A wave of the strip kernel that ran out of its poll budget flags the launch (its scores are wrong).  The tile
            // form, which computes the same values without waits between waves, is enqueued behind it and runs only if
            // the flag is set -- on the device: no copy back, no synchronisation, the call stays asynchronous -- and
            // counts itself (garlic_call_stats::n_stall_reruns: expected 0; a liveness bug shows there, not as a slow call)
            if (getenv("GARLIC_WLOD_STRIP_FORCE_RERUN")) HIP_TRY(hipMemsetAsync(p->d_counter.p + 3, 1, sizeof(int32_t), ctx->stream));
            WlodArgs ar = a;
            ar.run_if = p->d_counter.p + 3;
            ar.rerun_count = p->d_counter.p + 4;
            if (aligned16)
                hipLaunchKernelGGL((wlod_tile_glring_kernel<WLOD_R, true>), dim3(wl_grid), wl_block, wlod_lds, ctx->stream,
                                   a_packed, a_wtab, a_skew, d_out, ar);
            else
                hipLaunchKernelGGL((wlod_tile_glring_kernel<WLOD_R, false>), dim3(wl_grid), wl_block, wlod_lds, ctx->stream,
                                   a_packed, a_wtab, a_skew, d_out, ar);
        } else if (wlod_gl_ring && aligned16)
            hipLaunchKernelGGL((wlod_tile_glring_kernel<WLOD_R, true>), dim3(wl_grid), wl_block, wlod_lds, ctx->stream,
                               a_packed, a_wtab, a_skew, d_out, a);
        else if (wlod_gl_ring)
            hipLaunchKernelGGL((wlod_tile_glring_kernel<WLOD_R, false>), dim3(wl_grid), wl_block, wlod_lds, ctx->stream,
                               a_packed, a_wtab, a_skew, d_out, a);
        else if (wlod_gl && aligned16)
            hipLaunchKernelGGL((wlod_tile_gl_kernel<WLOD_R, true>), dim3(wl_grid), wl_block, wlod_lds, ctx->stream,
                               a_packed, a_wtab, a_skew, d_out, a);
        else if (wlod_gl)
            hipLaunchKernelGGL((wlod_tile_gl_kernel<WLOD_R, false>), dim3(wl_grid), wl_block, wlod_lds, ctx->stream,
                               a_packed, a_wtab, a_skew, d_out, a);
        else if (two_blocks && aligned16)
            hipLaunchKernelGGL((wlod_tile2_kernel<WLOD_R, true>), dim3(wl_grid), wl_block, wlod_lds, ctx->stream,
                               a_packed, a_wtab, a_skew, d_out, a);
        else if (two_blocks)
            hipLaunchKernelGGL((wlod_tile2_kernel<WLOD_R, false>), dim3(wl_grid), wl_block, wlod_lds, ctx->stream,
                               a_packed, a_wtab, a_skew, d_out, a);
        else if (aligned16)
            hipLaunchKernelGGL((wlod_tile_kernel<WLOD_R, true>), dim3(wl_grid), wl_block, wlod_lds, ctx->stream,
                               a_packed, a_wtab, a_skew, d_out, a);
        else
            hipLaunchKernelGGL((wlod_tile_kernel<WLOD_R, false>), dim3(wl_grid), wl_block, wlod_lds, ctx->stream,
                               a_packed, a_wtab, a_skew, d_out, a);
    } else if (n_items && exact) {
        VariantArgs a{p->d_packed.p, p->d_tab.p,  p->d_tabgl.p, p->d_codes.p, p->d_decay.p, p->d_rld.p,
                      p->d_items.p,  p->d_chrs.p, d_out,        p->nind_pad,  p->nwordrows, ind_begin,    ind_count,
                      W,             (int32_t)p->gl_values.size(), use_gl ? 1 : 0,
                      (use_gl && p->gl_cont) ? p->d_glterms.p : nullptr, (int64_t)(GOFF + p->nloci + GPAD_BACK)};
        hipLaunchKernelGGL(lod_chain_exact_kernel, dim3((unsigned)n_items), dim3(WAVE), 0, ctx->stream, a, (int)n_items);
    } else if (n_items && mode == MODE_LOD) {
        ChainArgs a{p->d_packed.p, p->d_tab.p, p->d_items.p,     p->d_chrs.p,     d_out, p->nind_pad, p->nwordrows,
                    ind_begin,     ind_count,  W,               (int32_t)n_items, p->d_counter.p, nullptr};
        DevBuf<int64_t> d_trace;   // debugging aid: GARLIC_TRACE=<file> dumps per-item timestamps
        const char *trace_path = getenv("GARLIC_TRACE");
        if (trace_path && !(feed_kernel && n_feed_items) && d_trace.reserve(4 * n_items) == GARLIC_OK) {
            (void)hipMemsetAsync(d_trace.p, 0, sizeof(int64_t) * 4 * n_items, ctx->stream);
            a.trace = d_trace.p;
        }
        if (feed_kernel && n_feed_items) {
            FeedArgs f{p->d_packed.p, p->d_tab.p, p->d_feed_items.p, p->d_chrs.p, d_out, nullptr, p->nwordrows, ind_begin, ind_count, W,
                       (int32_t)n_feed_items, thin_step, getenv("GARLIC_FEED_NO_ASM") ? 0 : 1, p->d_counter.p, nullptr};
            DevBuf<int64_t> d_ftrace;   // debugging aid: GARLIC_TRACE=<file> dumps per-item time stamps
            const char *ftrace_path = getenv("GARLIC_TRACE");
            if (ftrace_path && d_ftrace.reserve(8 * n_feed_items) == GARLIC_OK) {
                (void)hipMemsetAsync(d_ftrace.p, 0, sizeof(int64_t) * 8 * n_feed_items, ctx->stream);
                f.trace = d_ftrace.p;
            }
            const void *fn = (const void *)lod_feed_kernel;
            int grid = (int)std::min<size_t>(n_feed_items, (size_t)ctx->n_cu * std::max(1, p->plan.feed_per_cu));
            if (!reuse && (rc = feed_grid(ctx, feed_items, &grid, &p->plan.feed_per_cu))) return rc;
            void *kargs[] = {(void *)&f};
            HIP_TRY(hipLaunchKernel(fn, dim3((unsigned)grid), dim3(FEED_G * WAVE), kargs, 0, ctx->stream));
            if (f.trace) {
                std::vector<int64_t> tr(8 * n_feed_items);
                (void)hipMemcpyAsync(tr.data(), d_ftrace.p, sizeof(int64_t) * tr.size(), hipMemcpyDeviceToHost, ctx->stream);
                (void)hipStreamSynchronize(ctx->stream);
                if (FILE *fo = fopen(ftrace_path, "w")) {
                    for (size_t i = 0; i < n_feed_items; i++) {
                        fprintf(fo, "%zu", i);
                        for (int q = 0; q < 8; q++) fprintf(fo, " %lld", (long long)tr[8 * i + q]);
                        fprintf(fo, "\n");
                    }
                    fclose(fo);
                }
                d_ftrace.release();
            }
        } else if (aligned16)
            hipLaunchKernelGGL((lod_chain_kernel<true>), dim3((unsigned)workers), dim3(CHAIN_THREADS), 0,
                               ctx->stream, a);
        else
            hipLaunchKernelGGL((lod_chain_kernel<false>), dim3((unsigned)workers), dim3(CHAIN_THREADS), 0,
                               ctx->stream, a);
        if (a.trace) {
            std::vector<int64_t> tr(4 * n_items);
            (void)hipMemcpyAsync(tr.data(), d_trace.p, sizeof(int64_t) * tr.size(), hipMemcpyDeviceToHost, ctx->stream);
            (void)hipStreamSynchronize(ctx->stream);
            if (FILE *f = fopen(trace_path, "w")) {
                for (size_t i = 0; i < n_items; i++)
                    fprintf(f, "%zu %lld %lld %lld %lld %d\n", i, (long long)tr[4 * i], (long long)tr[4 * i + 1],
                            (long long)tr[4 * i + 2], (long long)tr[4 * i + 3], 0);
                fclose(f);
            }
            d_trace.release();
        }
    } else if (n_items) {
        VariantArgs a{p->d_packed.p, p->d_tab.p,  p->d_tabgl.p, p->d_codes.p, p->d_decay.p, p->d_rld.p,
                      p->d_items.p,  p->d_chrs.p, d_out,        p->nind_pad,  p->nwordrows, ind_begin,    ind_count,
                      W,             (int32_t)p->gl_values.size(), use_gl ? 1 : 0,
                      (use_gl && p->gl_cont) ? p->d_glterms.p : nullptr, (int64_t)(GOFF + p->nloci + GPAD_BACK)};
        if (mode == MODE_LOD_GL && p->glterms_valid && !p->glterms_scaled && (ind_begin & (WAVE - 1)) == 0 &&
            !getenv("GARLIC_TGLS_NO_RING")) {
            // persistent workgroups, every term row through an LDS ring once (tgls_ring_kernel.hpp)
            TglsArgs t{p->d_glterms.p, (int64_t)(GOFF + p->nloci + GPAD_BACK), p->d_items.p, p->d_chrs.p, d_out,
                       ind_begin, ind_count, W, (int32_t)n_items, p->d_counter.p, p->cov_pending};
            if (p->cov_pending.bits) p->cov_written = true;
            hipLaunchKernelGGL(lod_chain_ring_kernel, dim3((unsigned)workers), dim3(TG_THREADS), 0, ctx->stream, t);
        } else if (cov_bits) {
            return GARLIC_INTERNAL_NO_BITS;      // the TGLS ring chain / the tuned wLOD kernels do not take this shape
        } else if (mode == MODE_LOD_GL && p->glterms_valid && !p->glterms_scaled) {
            hipLaunchKernelGGL(lod_chain_terms_kernel, dim3((unsigned)n_items), dim3(2 * WAVE), 0, ctx->stream, a,
                               (int)n_items, (int64_t)(GOFF + p->nloci + GPAD_BACK), p->d_glterms.p);
        } else if (mode == MODE_LOD_GL) {
            hipLaunchKernelGGL(lod_chain_gl_kernel, dim3((unsigned)((n_items + GL_WAVES - 1) / GL_WAVES)),
                               dim3(GL_WAVES * WAVE), 0, ctx->stream, a, (int)n_items);
        } else {
            const int ring = W + TILE;
            const size_t lds = sizeof(double) * ((size_t)ring * WAVE + ((W + 1) & ~1) + (size_t)WAVE * TPITCH);
            if (lds > 160 * 1024)
                return fail(GARLIC_ERR_INVALID, "wLOD with winsize %d: this build supports 2..240 and 16..4096", W);
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(wlod_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(wlod_kernel, dim3((unsigned)n_items), dim3(WAVE), lds, ctx->stream, a,
                               ring);
        }
    }
    HIP_TRY(hipEventRecord(ctx->hist1[ctx->n_calls % garlic_ctx::HIST], ctx->stream));
    ctx->n_calls++;
    HIP_TRY(hipGetLastError());
    return GARLIC_OK;
    };   // enqueue
    if ((rc = enqueue())) return rc;
    p->last_chain_kind = exact ? 2 : 0;
    if (exact_possible && !exact && n_items) {
        int32_t found = 0;
        HIP_TRY(hipMemsetAsync(p->d_counter.p + 2, 0, sizeof(int32_t), ctx->stream));
        hipLaunchKernelGGL(sentinel_scan_kernel, dim3((unsigned)n_items), dim3(256), 0, ctx->stream, p->d_items.p, p->d_chrs.p,
                           d_out, ind_count, p->d_counter.p + 2);
        HIP_TRY(hipMemcpyAsync(&found, p->d_counter.p + 2, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        p->last_chain_kind = found ? 2 : 1;
        if (found) {
            exact = true;
            if ((rc = enqueue())) return rc;
        }
    }
    if (where == GARLIC_HOST)
        for (int c = 0; c < p->nchr; c++)
            HIP_TRY(hipMemcpy2DAsync(out + Lhost.base[c], sizeof(double) * Lhost.pitch[c], d_out + L.base[c],
                                     sizeof(double) * L.pitch[c], sizeof(double) * p->chr_nloci[c],
                                     (size_t)ind_count, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipEventRecord(ctx->ev_end, ctx->stream));
    // The work list lives in host vectors and per-panel device scratch: finish before returning --
    // unless the context is asynchronous, the output stays on the device and the plan (already
    // resident) is being reused: then nothing on the host is needed again and the call only enqueues.
    const bool enqueue_only = ctx->async_device && where == GARLIC_DEVICE && reuse;
    if (!enqueue_only) HIP_TRY(hipStreamSynchronize(ctx->stream));
    p->stats_pending = true;

    garlic_call_stats &st = p->stats;
    st.n_segments = (int64_t)p->boundaries.size();
    st.n_runs = n_runs;
    st.n_chain_items = (int64_t)n_items;
    p->plan.valid = true;
    p->plan.mode = (int)mode; p->plan.W = W; p->plan.max_gap = max_gap; p->plan.ind_begin = ind_begin;
    p->plan.ind_count = ind_count; p->plan.pitch_align = pitch_align;
    p->plan.wlod_fast = wlod_fast; p->plan.wlod_strip = wlod_gl_strip; p->plan.thin_step = thin_step; p->plan.blocks_hash = blocks_hash;
    p->plan.feed_kernel = feed_kernel; p->plan.n_feed_items = n_feed_items;
    p->plan.n_items = n_items; p->plan.n_fill = n_fill; p->plan.n_runs = n_runs; p->plan.n_valid = n_valid;
    st.n_valid_windows = n_valid;
    st.n_missing = p->nloci - n_valid;
    return GARLIC_OK;
}

} // namespace

// =================================================================================== C ABI
extern "C" {

int garlic_hip_abi_version(void) { return GARLIC_HIP_ABI_VERSION; }

const char *garlic_hip_last_error(void) { return g_last_error.c_str(); }

int garlic_hip_device_count(int32_t *count)
{
    if (!count) return fail(GARLIC_ERR_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(GARLIC_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return GARLIC_OK;
}

int garlic_ctx_create(int32_t device, void *hip_stream, garlic_ctx **out)
{
    if (!out) return fail(GARLIC_ERR_INVALID, "ctx out pointer is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n < 1)
        return fail(GARLIC_ERR_HIP, "no HIP device available (%s); libgarlic_hip has no CPU path",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= n)
        return fail(GARLIC_ERR_INVALID, "device %d out of range (have %d)", device, n);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(GARLIC_ERR_HIP, "device %d is %s; this library is built for gfx950 only", device,
                    prop.gcnArchName);
    garlic_ctx *ctx = new garlic_ctx;
    ctx->device = device;
    ctx->n_cu = std::max(1, prop.multiProcessorCount);   // a partitioned device (CPX / DPX) shows 32-128 of the 256
    if (hip_stream) {
        ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    } else {
        hipError_t se = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (se != hipSuccess) {
            delete ctx;
            return fail(GARLIC_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(se));
        }
        ctx->own_stream = true;
    }
    hipEvent_t *evs[2] = {&ctx->ev_begin, &ctx->ev_end};
    for (auto ev : evs) {
        hipError_t ee = hipEventCreate(ev);
        if (ee != hipSuccess) {
            delete ctx;
            return fail(GARLIC_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(ee));
        }
    }
    for (int i = 0; i < garlic_ctx::HIST; i++)
        if (hipEventCreate(&ctx->hist0[i]) != hipSuccess || hipEventCreate(&ctx->hist1[i]) != hipSuccess) {
            (void)garlic_ctx_destroy(ctx);
            return fail(GARLIC_ERR_HIP, "hipEventCreate failed");
        }
    *out = ctx;
    return GARLIC_OK;
}

int garlic_ctx_destroy(garlic_ctx *ctx)
{
    if (!ctx) return GARLIC_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (hipEvent_t ev : {ctx->ev_begin, ctx->ev_end})
        if (ev) (void)hipEventDestroy(ev);
    for (int i = 0; i < garlic_ctx::HIST; i++) {
        if (ctx->hist0[i]) (void)hipEventDestroy(ctx->hist0[i]);
        if (ctx->hist1[i]) (void)hipEventDestroy(ctx->hist1[i]);
    }
    ctx->d_logtab.release();
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return GARLIC_OK;
}

int garlic_ctx_synchronize(garlic_ctx *ctx)
{
    if (!ctx) return fail(GARLIC_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return GARLIC_OK;
}

int garlic_panel_create(garlic_ctx *ctx, int32_t nchr, const int32_t *chr_nloci, int32_t nind,
                        garlic_panel **out)
{
    if (!out) return fail(GARLIC_ERR_INVALID, "panel out pointer is NULL");
    *out = nullptr;
    if (!ctx) return fail(GARLIC_ERR_INVALID, "ctx is NULL");
    // initWinData refuses nind < 1 or nloci < 1 (src/garlic-data.cpp:1610-1620)
    if (nchr < 1 || !chr_nloci || nind < 1)
        return fail(GARLIC_ERR_INVALID, "need nchr >= 1, chr_nloci and nind >= 1");
    int rc;
    if ((rc = set_device(ctx))) return rc;
    garlic_panel *p = new garlic_panel;
    p->ctx = ctx;
    p->nchr = nchr;
    p->nind = nind;
    p->chr_nloci.assign(chr_nloci, chr_nloci + nchr);
    p->chr_off.resize(nchr + 1);
    p->chr_off[0] = 0;
    for (int c = 0; c < nchr; c++) {
        if (chr_nloci[c] < 1) {
            delete p;
            return fail(GARLIC_ERR_INVALID, "chromosome %d has %d loci; must be positive", c,
                        chr_nloci[c]);
        }
        p->chr_off[c + 1] = p->chr_off[c] + chr_nloci[c];
    }
    p->nloci = p->chr_off[nchr];
    p->nind_pad = ((int64_t)nind + 63 + 63) / 64 * 64;
    p->nwordrows = ((((GOFF + p->nloci + GPAD_BACK) >> 4) + 2) + 15) & ~(int64_t)15; // whole 4 KB chunks
    auto cleanup = [&](int code) { garlic_panel_destroy(p); return code; };
    if ((rc = p->d_packed.reserve((size_t)(p->nwordrows * p->nind_pad)))) return cleanup(rc);
    if ((rc = p->d_chr_off.reserve(nchr + 1))) return cleanup(rc);
    // every 2-bit code starts out "missing" (3): pad rows/columns contribute +0.0 and are never stored
    hipLaunchKernelGGL(fill_u32_kernel, dim3(2048), dim3(256), 0, ctx->stream, p->d_packed.p,
                       p->nwordrows * p->nind_pad, 0xFFFFFFFFu);
    hipError_t e = hipMemcpyAsync(p->d_chr_off.p, p->chr_off.data(), sizeof(int64_t) * (nchr + 1),
                                  hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess)
        return cleanup(fail(GARLIC_ERR_HIP, "panel init: %s", hipGetErrorString(e)));
    *out = p;
    return GARLIC_OK;
}

static void release_feed_slots(garlic_panel *p)
{
    for (auto *sl : p->feed_slots) {
        if (sl->stream) { (void)hipStreamSynchronize(sl->stream); (void)hipStreamDestroy(sl->stream); }
        if (sl->ev0) (void)hipEventDestroy(sl->ev0);
        if (sl->ev1) (void)hipEventDestroy(sl->ev1);
        sl->items.release(); sl->chrs.release(); sl->counter.release(); sl->row_counts.release(); sl->out.release(); sl->feed.release();
        delete sl;
    }
    p->feed_slots.clear();
}

int garlic_panel_destroy(garlic_panel *p)
{
    if (!p) return GARLIC_OK;
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    p->d_packed.release(); p->d_pos.release(); p->d_cs.release(); p->d_ce.release();
    p->d_chr_off.release(); p->d_tab.release(); p->d_blk_counts.release();
    p->d_blk_offsets.release(); p->d_total.release(); p->d_boundaries.release();
    p->d_items.release(); p->d_fill.release(); p->d_counter.release(); p->d_chrs.release(); p->d_stage16.release(); p->d_row_counts.release(); p->d_codes.release(); p->d_tabgl.release();
    p->d_rld.release(); p->d_decay.release(); p->d_stage64.release(); p->d_phase.release(); p->lds.release();
    p->d_glterms.release(); p->d_glval.release(); p->d_freq.release(); p->d_skew.release(); p->d_wtab.release(); p->d_valid.release(); p->d_tiles.release(); p->d_segs.release(); p->d_strips.release();
    p->d_out.release(); p->d_feed.release(); p->d_feed_items.release();
    release_feed_slots(p);
    delete p;
    return GARLIC_OK;
}

int garlic_panel_set_map(garlic_panel *p, const int32_t *pos, const double *gpos,
                         const int32_t *centro_start, const int32_t *centro_end)
{
    if (!p || !pos || !centro_start || !centro_end)
        return fail(GARLIC_ERR_INVALID, "panel, pos, centro_start and centro_end are required");
    int rc;
    if ((rc = set_device(p->ctx))) return rc;
    p->pos.assign(pos, pos + p->nloci);
    p->cs.assign(centro_start, centro_start + p->nchr);
    p->ce.assign(centro_end, centro_end + p->nchr);
    p->have_gpos = gpos != nullptr;
    if (gpos) p->gpos.assign(gpos, gpos + p->nloci);
    if ((rc = p->d_pos.reserve((size_t)p->nloci))) return rc;
    if ((rc = p->d_cs.reserve(p->nchr))) return rc;
    if ((rc = p->d_ce.reserve(p->nchr))) return rc;
    hipStream_t s = p->ctx->stream;
    HIP_TRY(hipMemcpyAsync(p->d_pos.p, pos, sizeof(int32_t) * p->nloci, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(p->d_cs.p, centro_start, sizeof(int32_t) * p->nchr, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(p->d_ce.p, centro_end, sizeof(int32_t) * p->nchr, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    p->have_map = true;
    p->seg_valid = false;
    p->decay_valid = false;
    p->plan.valid = false;
    if (p->glterms_scaled) p->glterms_valid = false;   // (term * nomut) * norec of the old positions
    return GARLIC_OK;
}

int garlic_panel_set_freq(garlic_panel *p, const double *freq)
{
    if (!p || !freq) return fail(GARLIC_ERR_INVALID, "panel and freq are required");
    p->freq.assign(freq, freq + p->nloci);
    p->have_freq = true;
    p->tab_valid = false;
    p->tabgl_valid = false;
    p->glterms_valid = false;
    p->dfreq_valid = false;
    return GARLIC_OK;
}

int garlic_panel_set_genotypes(garlic_panel *p, const int16_t *geno, int64_t ld, int64_t locus_begin,
                               int64_t locus_count, int32_t where)
{
    if (!p || !geno) return fail(GARLIC_ERR_INVALID, "panel and geno are required");
    if (ld < p->nind) return fail(GARLIC_ERR_INVALID, "ld %lld < nind %d", (long long)ld, p->nind);
    if (locus_begin < 0 || locus_count < 1 || locus_begin + locus_count > p->nloci)
        return fail(GARLIC_ERR_INVALID, "locus range [%lld,+%lld) outside panel of %lld loci",
                    (long long)locus_begin, (long long)locus_count, (long long)p->nloci);
    int rc;
    if ((rc = set_device(p->ctx))) return rc;
    hipStream_t s = p->ctx->stream;
    // host data goes through a bounded staging buffer, a slab of SNP rows at a time
    const int64_t slab_rows = (where == GARLIC_HOST)
                                  ? std::max<int64_t>(16, ((int64_t)256 << 20) / (2 * ld))
                                  : locus_count;
    for (int64_t done = 0; done < locus_count; done += slab_rows) {
        const int64_t rows = std::min(slab_rows, locus_count - done);
        const int64_t l0 = locus_begin + done;
        const int16_t *src = geno + done * ld;
        if (where == GARLIC_HOST) {
            if ((rc = p->d_stage16.reserve((size_t)(rows * ld)))) return rc;
            HIP_TRY(hipMemcpyAsync(p->d_stage16.p, src, sizeof(int16_t) * rows * ld,
                                   hipMemcpyHostToDevice, s));
            src = p->d_stage16.p;
        }
        const int64_t w_lo = (GOFF + l0) >> 4;
        const int64_t w_hi = ((GOFF + l0 + rows - 1) >> 4) + 1;
        dim3 block(256);
        for (int64_t w = w_lo; w < w_hi; w += 65535) {
            const int64_t wn = std::min<int64_t>(65535, w_hi - w);
            dim3 grid((unsigned)((p->nind_pad + 255) / 256), (unsigned)wn);
            hipLaunchKernelGGL(pack_genotypes_kernel, grid, block, 0, s, src, ld, l0, rows, p->nind,
                               p->nind_pad, p->nwordrows, p->d_packed.p, w, w + wn);
        }
        if (where == GARLIC_HOST) HIP_TRY(hipStreamSynchronize(s)); // staging buffer is reused
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    p->have_geno = true;
    p->geno_epoch++;
    p->glterms_valid = false;
    return GARLIC_OK;
}

int garlic_panel_set_genotypes_2bit(garlic_panel *p, const uint8_t *rows, int64_t row_bytes, int64_t ind_offset,
                                    int64_t locus_begin, int64_t locus_count, int32_t where)
{
    if (!p || !rows) return fail(GARLIC_ERR_INVALID, "panel and rows are required");
    if (ind_offset < 0 || row_bytes < (ind_offset + p->nind + 3) / 4)
        return fail(GARLIC_ERR_INVALID, "row_bytes %lld too small for individuals [%lld,+%d)", (long long)row_bytes,
                    (long long)ind_offset, p->nind);
    if (locus_begin < 0 || locus_count < 1 || locus_begin + locus_count > p->nloci)
        return fail(GARLIC_ERR_INVALID, "locus range [%lld,+%lld) outside panel of %lld loci",
                    (long long)locus_begin, (long long)locus_count, (long long)p->nloci);
    int rc;
    if ((rc = set_device(p->ctx))) return rc;
    hipStream_t s = p->ctx->stream;
    DevBuf<uint8_t> stage;
    const int64_t slab_rows = (where == GARLIC_HOST) ? std::max<int64_t>(16, ((int64_t)256 << 20) / row_bytes)
                                                     : locus_count;
    for (int64_t done = 0; done < locus_count; done += slab_rows) {
        const int64_t nrows = std::min(slab_rows, locus_count - done);
        const int64_t l0 = locus_begin + done;
        const uint8_t *src = rows + done * row_bytes;
        hipError_t e = hipSuccess;
        if (where == GARLIC_HOST) {
            if ((rc = stage.reserve((size_t)(nrows * row_bytes)))) { stage.release(); return rc; }
            e = hipMemcpyAsync(stage.p, src, (size_t)(nrows * row_bytes), hipMemcpyHostToDevice, s);
            src = stage.p;
        }
        const int64_t w_lo = (GOFF + l0) >> 4;
        const int64_t w_hi = ((GOFF + l0 + nrows - 1) >> 4) + 1;
        for (int64_t w = w_lo; e == hipSuccess && w < w_hi; w += 65535) {
            const int64_t wn = std::min<int64_t>(65535, w_hi - w);
            dim3 grid((unsigned)((p->nind_pad + 255) / 256), (unsigned)wn);
            hipLaunchKernelGGL(pack_genotypes_2bit_kernel, grid, dim3(256), 0, s, src, row_bytes, ind_offset, l0, nrows,
                               p->nind, p->nind_pad, p->nwordrows, p->d_packed.p, w, w + wn);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(s);                      // staging buffer is reused
        if (e != hipSuccess) { stage.release(); return fail(GARLIC_ERR_HIP, "set_genotypes_2bit: %s", hipGetErrorString(e)); }
    }
    stage.release();
    p->have_geno = true;
    p->geno_epoch++;
    p->glterms_valid = false;
    return GARLIC_OK;
}

int garlic_panel_set_gl(garlic_panel *p, const double *gl, int64_t ld, int64_t locus_begin,
                        int64_t locus_count, int32_t where)
{
    if (!p || !gl) return fail(GARLIC_ERR_INVALID, "panel and gl are required");
    if (ld < p->nind) return fail(GARLIC_ERR_INVALID, "ld %lld < nind %d", (long long)ld, p->nind);
    if (locus_begin < 0 || locus_count < 1 || locus_begin + locus_count > p->nloci)
        return fail(GARLIC_ERR_INVALID, "locus range [%lld,+%lld) outside panel of %lld loci",
                    (long long)locus_begin, (long long)locus_count, (long long)p->nloci);
    int rc;
    if ((rc = set_device(p->ctx))) return rc;
    hipStream_t s = p->ctx->stream;
    const int64_t rows_total = GOFF + p->nloci + GPAD_BACK;
    if ((rc = restart_continuous_upload(p))) return rc;
    if (!p->gl_cont && getenv("GARLIC_TGLS_CONTINUOUS") && (rc = switch_to_continuous(p))) return rc;
    if (!p->gl_cont && !p->d_codes.p) {
        if ((rc = p->d_codes.reserve((size_t)(rows_total * p->nind_pad)))) return rc;
        HIP_TRY(hipMemsetAsync(p->d_codes.p, 0, (size_t)(rows_total * p->nind_pad), s));
    }
    // Few distinct error probabilities (GQ / PL integers) -> one-byte codes: the term table per (SNP,
    // code, genotype) comes from the host libm and the panel keeps 1 B instead of 8 B per genotype.
    // Coded on the device against the dictionary so far; values it does not know come back, join the
    // dictionary and the slab is coded again (a hash look-up per genotype on the host took minutes at
    // 1e10 genotypes).  When the dictionary is full (256 values: --gl-type GL, continuous inputs) the
    // panel switches to keeping the values themselves and evaluates lod() on the device.
    constexpr int UNK_CAP = 8192;
    DevBuf<double> stage;
    DevBuf<uint64_t> d_bits, d_unk;
    DevBuf<uint8_t> d_dcode;
    DevBuf<int32_t> d_nunk;
    auto done = [&](int code) { stage.release(); d_bits.release(); d_unk.release(); d_dcode.release(); d_nunk.release(); return code; };
    if ((rc = d_bits.reserve(GL_DICT_MAX)) || (rc = d_dcode.reserve(GL_DICT_MAX)) || (rc = d_unk.reserve(UNK_CAP)) ||
        (rc = d_nunk.reserve(1)))
        return done(rc);
    const int64_t slab_rows = (where == GARLIC_HOST) ? std::max<int64_t>(16, ((int64_t)256 << 20) / (8 * ld)) : locus_count;
    std::vector<uint64_t> unk(UNK_CAP);
    for (int64_t at = 0; at < locus_count; at += slab_rows) {
        const int64_t nrows = std::min(slab_rows, locus_count - at);
        const double *src = gl + at * ld;
        hipError_t e = hipSuccess;
        if (where == GARLIC_HOST) {
            if ((rc = stage.reserve((size_t)(nrows * ld)))) return done(rc);
            e = hipMemcpyAsync(stage.p, src, sizeof(double) * nrows * ld, hipMemcpyHostToDevice, s);
            src = stage.p;
        }
        while (e == hipSuccess && !p->gl_cont) {   // until the slab holds no value outside the dictionary
            // dictionary, sorted by bit pattern
            std::vector<std::pair<uint64_t, uint8_t>> dict;
            for (auto &kv : p->gl_code) dict.emplace_back(kv.first, (uint8_t)kv.second);
            std::sort(dict.begin(), dict.end());
            std::vector<uint64_t> hb(dict.size());
            std::vector<uint8_t> hc(dict.size());
            for (size_t k = 0; k < dict.size(); k++) { hb[k] = dict[k].first; hc[k] = dict[k].second; }
            if (!dict.empty()) {
                e = hipMemcpyAsync(d_bits.p, hb.data(), sizeof(uint64_t) * hb.size(), hipMemcpyHostToDevice, s);
                if (e == hipSuccess) e = hipMemcpyAsync(d_dcode.p, hc.data(), hc.size(), hipMemcpyHostToDevice, s);
            }
            if (e == hipSuccess) e = hipMemsetAsync(d_nunk.p, 0, sizeof(int32_t), s);
            if (e != hipSuccess) break;
            hipLaunchKernelGGL(gl_encode_kernel, dim3(2048), dim3(256), 0, s, src, ld, nrows, p->nind, p->nind_pad, d_bits.p,
                               d_dcode.p, (int)dict.size(), p->d_codes.p + (GOFF + locus_begin + at) * p->nind_pad,
                               d_unk.p, d_nunk.p, UNK_CAP);
            int32_t nunk = 0;
            e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpyAsync(&nunk, d_nunk.p, sizeof nunk, hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);          // also: hb / hc / the staging slab are free again
            if (e != hipSuccess || nunk == 0) break;
            const int got = std::min<int32_t>(nunk, UNK_CAP);
            e = hipMemcpy(unk.data(), d_unk.p, sizeof(uint64_t) * got, hipMemcpyDeviceToHost);
            if (e != hipSuccess) break;
            for (int k = 0; k < got; k++) {
                if (p->gl_code.count(unk[k])) continue;
                const int code = (int)p->gl_values.size();
                if (code >= GL_DICT_MAX) {       // continuous inputs: keep values, not codes
                    if ((rc = switch_to_continuous(p))) return done(rc);
                    break;
                }
                double v;
                memcpy(&v, &unk[k], sizeof v);
                p->gl_code.emplace(unk[k], code);
                p->gl_values.push_back(v);
                p->tabgl_valid = false;
            }
        }
        if (e == hipSuccess && p->gl_cont) {
            hipLaunchKernelGGL(gl_store_kernel, dim3(2048), dim3(256), 0, s, src, ld, locus_begin + at, nrows, p->nind,
                               rows_total, p->d_glval.p);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(s);          // the staging slab is free again
        }
        if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "set_gl: %s", hipGetErrorString(e)));
    }
    if (p->gl_cont && !p->gl_cover.empty())
        memset(p->gl_cover.data() + locus_begin, 1, (size_t)locus_count);
    p->have_gl = true;
    p->glterms_valid = false;
    return done(GARLIC_OK);
}

int garlic_panel_set_gl_codes(garlic_panel *p, const uint8_t *codes, int64_t ld, int64_t locus_begin,
                              int64_t locus_count, const double *values, int32_t nvalues, int32_t where)
{
    if (!p || !codes || !values) return fail(GARLIC_ERR_INVALID, "panel, codes and values are required");
    if (nvalues < 1 || nvalues > GL_DICT_MAX) return fail(GARLIC_ERR_INVALID, "nvalues must be 1..256 (got %d)", nvalues);
    if (ld < p->nind) return fail(GARLIC_ERR_INVALID, "ld %lld < nind %d", (long long)ld, p->nind);
    if (locus_begin < 0 || locus_count < 1 || locus_begin + locus_count > p->nloci)
        return fail(GARLIC_ERR_INVALID, "locus range [%lld,+%lld) outside panel of %lld loci",
                    (long long)locus_begin, (long long)locus_count, (long long)p->nloci);
    int rc;
    if ((rc = set_device(p->ctx))) return rc;
    hipStream_t s = p->ctx->stream;
    const int64_t rows_total = GOFF + p->nloci + GPAD_BACK;
    if ((rc = restart_continuous_upload(p))) return rc;
    if (!p->gl_cont && getenv("GARLIC_TGLS_CONTINUOUS") && (rc = switch_to_continuous(p))) return rc;
    // the caller's table joins the panel's dictionary; its codes are translated on the device.  A
    // panel whose tables add up to more than 256 values keeps the values themselves from then on.
    uint8_t remap[256] = {0};
    for (int k = 0; k < nvalues && !p->gl_cont; k++) {
        uint64_t bits;
        memcpy(&bits, &values[k], sizeof bits);
        auto it = p->gl_code.find(bits);
        if (it == p->gl_code.end()) {
            const int code = (int)p->gl_values.size();
            if (code >= GL_DICT_MAX) {
                if (!p->d_codes.p) {   // nothing coded yet: start from an empty value matrix
                    p->gl_code.clear();
                    p->gl_values.clear();
                }
                if ((rc = switch_to_continuous(p))) return rc;
                break;
            }
            it = p->gl_code.emplace(bits, code).first;
            p->gl_values.push_back(values[k]);
            p->tabgl_valid = false;
        }
        remap[k] = (uint8_t)it->second;
    }
    if (!p->gl_cont && !p->d_codes.p) {
        if ((rc = p->d_codes.reserve((size_t)(rows_total * p->nind_pad)))) return rc;
        HIP_TRY(hipMemsetAsync(p->d_codes.p, 0, (size_t)(rows_total * p->nind_pad), s));
    }
    DevBuf<uint8_t> stage, d_remap;
    DevBuf<double> d_dict;
    auto done = [&](int code) { stage.release(); d_remap.release(); d_dict.release(); return code; };
    hipError_t e = hipSuccess;
    if (p->gl_cont) {
        std::vector<double> dict(GL_DICT_MAX, 0.0);
        std::copy(values, values + nvalues, dict.begin());
        if ((rc = d_dict.reserve(GL_DICT_MAX))) return done(rc);
        e = hipMemcpy(d_dict.p, dict.data(), sizeof(double) * GL_DICT_MAX, hipMemcpyHostToDevice);
    } else {
        if ((rc = d_remap.reserve(256))) return done(rc);
        e = hipMemcpy(d_remap.p, remap, 256, hipMemcpyHostToDevice);
    }
    const int64_t slab_rows = (where == GARLIC_HOST) ? std::max<int64_t>(16, ((int64_t)256 << 20) / ld) : locus_count;
    for (int64_t at = 0; e == hipSuccess && at < locus_count; at += slab_rows) {
        const int64_t nrows = std::min(slab_rows, locus_count - at);
        const uint8_t *src = codes + at * ld;
        if (where == GARLIC_HOST) {
            if ((rc = stage.reserve((size_t)(nrows * ld)))) return done(rc);
            e = hipMemcpyAsync(stage.p, src, (size_t)(nrows * ld), hipMemcpyHostToDevice, s);
            src = stage.p;
        }
        if (e != hipSuccess) break;
        if (p->gl_cont)
            hipLaunchKernelGGL(gl_store_codes_kernel, dim3(2048), dim3(256), 0, s, src, ld, locus_begin + at, nrows, p->nind,
                               d_dict.p, rows_total, p->d_glval.p);
        else
            hipLaunchKernelGGL(gl_recode_kernel, dim3(2048), dim3(256), 0, s, src, ld, nrows, p->nind, p->nind_pad, d_remap.p,
                               p->d_codes.p + (GOFF + locus_begin + at) * p->nind_pad);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(s);                      // staging slab free again
    }
    if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "set_gl_codes: %s", hipGetErrorString(e)));
    if (p->gl_cont && !p->gl_cover.empty())
        memset(p->gl_cover.data() + locus_begin, 1, (size_t)locus_count);
    p->have_gl = true;
    p->glterms_valid = false;
    return done(GARLIC_OK);
}

int garlic_panel_set_phase(garlic_panel *p, const uint8_t *first_copy, int64_t ld, int64_t locus_begin,
                           int64_t locus_count, int32_t where)
{
    if (!p || !first_copy) return fail(GARLIC_ERR_INVALID, "panel and first_copy are required");
    if (ld < p->nind) return fail(GARLIC_ERR_INVALID, "ld %lld < nind %d", (long long)ld, p->nind);
    if (locus_begin < 0 || locus_count < 1 || locus_begin + locus_count > p->nloci)
        return fail(GARLIC_ERR_INVALID, "locus range [%lld,+%lld) outside panel of %lld loci",
                    (long long)locus_begin, (long long)locus_count, (long long)p->nloci);
    int rc;
    if ((rc = set_device(p->ctx))) return rc;
    hipStream_t s = p->ctx->stream;
    const int nblk = (int)(p->nind_pad / WAVE);
    if (!p->d_phase.p) {
        if ((rc = p->d_phase.reserve((size_t)nblk * p->nloci))) return rc;
        HIP_TRY(hipMemsetAsync(p->d_phase.p, 0, sizeof(uint64_t) * nblk * p->nloci, s));
    }
    DevBuf<uint8_t> stage;
    const int64_t slab_rows = (where == GARLIC_HOST) ? std::max<int64_t>(16, ((int64_t)256 << 20) / ld)
                                                     : locus_count;
    for (int64_t done = 0; done < locus_count; done += slab_rows) {
        const int64_t rows = std::min(slab_rows, locus_count - done);
        const uint8_t *src = first_copy + done * ld;
        hipError_t e = hipSuccess;
        if (where == GARLIC_HOST) {
            if ((rc = stage.reserve((size_t)(rows * ld)))) { stage.release(); return rc; }
            e = hipMemcpyAsync(stage.p, src, (size_t)(rows * ld), hipMemcpyHostToDevice, s);
            src = stage.p;
        }
        if (e == hipSuccess) {
            const int64_t waves = rows * nblk;
            hipLaunchKernelGGL(phase_planes_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, src,
                               ld, locus_begin + done, rows, p->nind, nblk, p->nloci, p->d_phase.p);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(s);                      // staging buffer is reused
        if (e != hipSuccess) { stage.release(); return fail(GARLIC_ERR_HIP, "set_phase: %s", hipGetErrorString(e)); }
    }
    stage.release();
    p->have_phase = true;
    return GARLIC_OK;
}

// reciprocals of device-resident LD weights, plain and skewed (what the wLOD kernels read)
// the skewed reciprocals D[l][j] = 1 / LD[l - j][j]: the weights SNP l has in the windows that contain it as one
// contiguous row (tuned wLOD kernels); rows past the panel stay 0 (SKEW_FRONT doubles of zero padding in front:
// the kernels' first steps read up to 15 elements before a row)
// zero: clear it (skew_reciprocal_kernel's caller).  ld_sum_col_kernel writes every weight a scored window reads; what it
// leaves alone only ever reaches windows that have no score (they are computed along and written as MISSING), so its
// caller clears just the padding behind the panel, which the last windows' loads run into.
static int reserve_skew(garlic_panel *p, int32_t winsize, bool zero = true)
{
    int rc;
    const size_t nskew = SKEW_FRONT + ((size_t)p->nloci + winsize + 64) * winsize;
    if ((rc = p->d_skew.reserve(nskew))) return rc;
    if (zero) {
        HIP_TRY(hipMemsetAsync(p->d_skew.p, 0, sizeof(double) * nskew, p->ctx->stream));
    } else {
        HIP_TRY(hipMemsetAsync(p->d_skew.p, 0, sizeof(double) * SKEW_FRONT, p->ctx->stream));
        const size_t body = SKEW_FRONT + (size_t)p->nloci * winsize;
        HIP_TRY(hipMemsetAsync(p->d_skew.p + body, 0, sizeof(double) * (nskew - body), p->ctx->stream));
    }
    return GARLIC_OK;
}

// skew_done: the LD kernels have written D themselves (ld_sum_col_kernel)
static int install_ld(garlic_panel *p, int32_t winsize, const double *src, bool skew_done = false)
{
    int rc;
    hipStream_t s = p->ctx->stream;
    if (!skew_done) {
        if ((rc = reserve_skew(p, winsize))) return rc;
        for (int c = 0; c < p->nchr; c++)
            hipLaunchKernelGGL(skew_reciprocal_kernel, dim3(1024), dim3(256), 0, s, src,
                               p->d_skew.p + SKEW_FRONT, p->chr_off[c], p->chr_off[c + 1], winsize);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    p->d_rld.release();          // the plain reciprocals (generic wLOD kernel) are made from D when they are asked for
    p->rld_valid = false;
    p->have_ld = true;
    p->ld_winsize = winsize;
    return GARLIC_OK;
}

static int ensure_rld(garlic_panel *p)
{
    if (p->rld_valid) return GARLIC_OK;
    int rc;
    const int32_t W = p->ld_winsize;
    if ((rc = p->d_rld.reserve((size_t)p->nloci * W))) return rc;
    for (int c = 0; c < p->nchr; c++)
        hipLaunchKernelGGL(unskew_kernel, dim3(1024), dim3(256), 0, p->ctx->stream, p->d_skew.p + SKEW_FRONT, p->d_rld.p,
                           p->chr_off[c], p->chr_off[c + 1], W);
    HIP_TRY(hipGetLastError());
    p->rld_valid = true;
    return GARLIC_OK;
}

int garlic_panel_set_ld(garlic_panel *p, int32_t winsize, const double *ld, int32_t where)
{
    if (!p || !ld) return fail(GARLIC_ERR_INVALID, "panel and ld are required");
    if (winsize <= 1) return fail(GARLIC_ERR_INVALID, "winsize must be > 1");
    int rc;
    if ((rc = set_device(p->ctx))) return rc;
    const size_t n = (size_t)p->nloci * winsize;
    const double *src = ld;
    if (where == GARLIC_HOST) {
        if ((rc = p->d_stage64.reserve(n))) return rc;
        HIP_TRY(hipMemcpyAsync(p->d_stage64.p, ld, sizeof(double) * n, hipMemcpyHostToDevice,
                               p->ctx->stream));
        src = p->d_stage64.p;
    }
    rc = install_ld(p, winsize, src);
    p->d_stage64.release();
    return rc;
}

// ---- LD weights on the device (ld_kernels.hpp)
// unphased, 16 < W <= 129: the pair counts as banded Gram matrices on the matrix cores (ld_pair_mfma_kernel)
static bool ld_pairs_on_mfma(int32_t winsize, int32_t phased)
{
    const int mfma_nj = 1 + (30 + winsize) / 32;
    const bool pair_flat = getenv("GARLIC_LD_PAIR_FLAT") && winsize <= 32;
    return !phased && !pair_flat && winsize > LD_SMALL_MAX_W && mfma_nj <= 5 && !getenv("GARLIC_LD_PAIR_NO_MFMA") &&
           !getenv("GARLIC_LD_PAIR_TILED") && !getenv("GARLIC_LD_PAIR_L2");
}
// 32 < W <= 512: the ordered sums with a thread per SNP of the window (ld_sum_col_kernel), from the combined hr2 table
static bool ld_sums_by_snp(int32_t winsize)
{
    const int col_threads = (winsize + 16 + WAVE - 1) / WAVE * WAVE;
    return winsize > LD_COL_B && winsize <= 512 && col_threads <= LD_COL_MAX_THREADS && !getenv("GARLIC_LD_SUM_BY_COLUMN") &&
           !getenv("GARLIC_LD_SUM_L2");
}

static int ld_check(garlic_panel *p, int32_t winsize, int32_t phased)
{
    if (!p) return fail(GARLIC_ERR_INVALID, "panel is NULL");
    if (!p->have_geno) return fail(GARLIC_ERR_STATE, "panel needs genotypes before LD weights");
    if (phased && !p->have_phase)
        return fail(GARLIC_ERR_STATE, "phased LD weights need garlic_panel_set_phase first");
    if (phased && !p->have_freq)
        return fail(GARLIC_ERR_STATE, "phased LD weights need the allele frequencies (garlic_panel_set_freq)");
    if (winsize <= 1) return fail(GARLIC_ERR_INVALID, "winsize must be > 1");
    if ((int64_t)p->nloci * winsize * 2 >= ((int64_t)1 << 40))
        return fail(GARLIC_ERR_INVALID, "LD table of %lld x %d too large", (long long)p->nloci, winsize);
    return set_device(p->ctx);
}

int garlic_ld_counts(garlic_panel *p, int32_t winsize, int32_t phased, const int32_t *sub_idx, int32_t n_sub,
                     int32_t *locus_counts, int32_t *pair_counts, int32_t where)
{
    int rc;
    if ((rc = ld_check(p, winsize, phased))) return rc;
    if (!locus_counts || !pair_counts) return fail(GARLIC_ERR_INVALID, "count buffers are required");
    if (n_sub < 0 || (n_sub > 0 && !sub_idx)) return fail(GARLIC_ERR_INVALID, "bad LD subsample");
    // sub_idx == NULL: every individual; otherwise exactly the n_sub listed ones -- none when n_sub is 0
    // (a shard that holds no member of a panel-wide subsample)
    hipStream_t s = p->ctx->stream;
    const int nblk = (int)(p->nind_pad / WAVE);
    // LD subsample as one bit per individual (order and repeats do not matter for counts of a set;
    // the reference draws distinct indices, garlic-data.cpp:361-362)
    std::vector<uint64_t> sub((size_t)nblk, 0);
    if (!sub_idx) {
        for (int i = 0; i < p->nind; i++) sub[i >> 6] |= (uint64_t)1 << (i & 63);
    } else {
        for (int k = 0; k < n_sub; k++) {
            const int i = sub_idx[k];
            if (i < 0 || i >= p->nind)
                return fail(GARLIC_ERR_INVALID, "LD subsample index %d outside panel of %d", i, p->nind);
            if (sub[i >> 6] & ((uint64_t)1 << (i & 63)))
                return fail(GARLIC_ERR_INVALID, "LD subsample index %d given twice", i);
            sub[i >> 6] |= (uint64_t)1 << (i & 63);
        }
    }
    DevBuf<uint64_t> &d_sub = p->lds.sub, &d_m = p->lds.m, &d_h = p->lds.h, &d_o = p->lds.o;
    DevBuf<int32_t> &d_loc = p->lds.loc, &d_pair = p->lds.pair;
    auto done = [&](int code) { return code; };   // the scratch stays with the panel
    const size_t npl = (size_t)nblk * p->nloci, npair = (size_t)p->nloci * winsize * 2;
    if ((rc = d_sub.reserve(nblk)) || (rc = d_m.reserve(npl)) || (rc = d_h.reserve(npl))) return done(rc);
    if (phased && (rc = d_o.reserve(npl))) return done(rc);
    int32_t *loc = locus_counts, *pair = pair_counts;
    if (where == GARLIC_HOST) {
        if ((rc = d_loc.reserve((size_t)p->nloci * 2)) || (rc = d_pair.reserve(npair))) return done(rc);
        loc = d_loc.p; pair = d_pair.p;
    }
    // pair counts: LDS-tiled (thread = distance, W - 1 <= 256; writes every entry of the table) or streamed from L2
    // pair counts: a lane per SNP while the tile's plane words fit LDS (ld_pair_lane_kernel); else a thread per distance
    // (ld_pair_tiled_kernel, W - 1 <= 256; writes every entry of the table too) or streamed from L2
    const bool pair_flat = getenv("GARLIC_LD_PAIR_FLAT") && winsize <= 32;
    const int lane_stage = std::min(nblk, getenv("GARLIC_LD_LANE_STAGE") ? atoi(getenv("GARLIC_LD_LANE_STAGE")) : 4);
    const int lane_dc = winsize - 1 <= 16 ? 16 : 32;
    const size_t lane_lds = sizeof(uint64_t) * (phased ? 4 : 2) * (size_t)lane_stage * (LD_LANE_T + winsize - 1);
    // unphased, 16 < W <= 129: the counts as banded Gram matrices on the matrix cores (ld_pair_mfma_kernel)
    const int mfma_nj = 1 + (30 + winsize) / 32;
    const bool pair_mfma = ld_pairs_on_mfma(winsize, phased);
    // garlic_panel_compute_ld (every individual's counts are local): the pair kernel writes the hr2 table itself
    const bool fuse_hr2 = p->lds.fuse_request && pair_mfma && ld_sums_by_snp(winsize);
    if (p->lds.fuse_request && !fuse_hr2) {      // (the caller has sized the pair table for the fused form)
        p->lds.fuse_request = false;
        return fail(GARLIC_ERR_STATE, "internal: LD fusion requested for a shape the pair kernel does not take");
    }
    p->lds.fuse_request = false;
    p->lds.fused_done = false;
    const bool pair_lane = !pair_mfma && !pair_flat && winsize - 1 <= 256 && lane_lds <= 150 * 1024 && !getenv("GARLIC_LD_PAIR_TILED") &&
                           !getenv("GARLIC_LD_PAIR_L2");
    const bool pair_tiled = !pair_mfma && !pair_flat && !pair_lane && winsize - 1 <= 256 && !getenv("GARLIC_LD_PAIR_L2");
    hipError_t e = hipMemcpyAsync(d_sub.p, sub.data(), sizeof(uint64_t) * nblk, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && !pair_tiled && !pair_flat && !pair_lane && !pair_mfma) e = hipMemsetAsync(pair, 0, sizeof(int32_t) * npair, s);
    if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "LD counts: %s", hipGetErrorString(e)));
    uint64_t planes_key = 0xCBF29CE484222325ull;
    for (uint64_t w : sub) planes_key = (planes_key ^ w) * 0x100000001B3ull;
    planes_key = (planes_key ^ (uint64_t)(phased ? 2 : 1)) * 0x100000001B3ull;
    planes_key = (planes_key ^ p->geno_epoch) * 0x100000001B3ull;
    planes_key = (planes_key ^ (uint64_t)nblk) * 0x100000001B3ull;
    if ((rc = p->lds.loc_planes.reserve((size_t)p->nloci * 2))) return done(rc);
    if (!(p->lds.planes_valid && p->lds.planes_key == planes_key) || getenv("GARLIC_LD_NO_PLANE_CACHE")) {
        p->lds.planes_valid = false;
        if (phased)
            hipLaunchKernelGGL(ld_planes_kernel<true>, dim3((unsigned)p->nwordrows), dim3(256), 0, s, p->d_packed.p,
                               p->nwordrows, nblk, d_sub.p, p->nloci, d_m.p, d_h.p, d_o.p, p->lds.loc_planes.p);
        else
            hipLaunchKernelGGL(ld_planes_kernel<false>, dim3((unsigned)p->nwordrows), dim3(256), 0, s, p->d_packed.p,
                               p->nwordrows, nblk, d_sub.p, p->nloci, d_m.p, d_h.p, (uint64_t *)nullptr, p->lds.loc_planes.p);
        e = hipGetLastError();
        if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "LD planes: %s", hipGetErrorString(e)));
        p->lds.planes_key = planes_key;
        p->lds.planes_valid = true;
    }
    e = hipMemcpyAsync(loc, p->lds.loc_planes.p, sizeof(int32_t) * p->nloci * 2, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "LD counts: %s", hipGetErrorString(e)));
    const int pair_threads = (winsize - 1 + WAVE - 1) / WAVE * WAVE;
    const size_t pair_lds = sizeof(uint64_t) * (phased ? 4 : 2) * LD_PAIR_BLK * (LD_PAIR_T + winsize - 1);
    if (pair_tiled && pair_lds > 48 * 1024) {
        const void *fn = phased ? (const void *)ld_pair_tiled_kernel<true> : (const void *)ld_pair_tiled_kernel<false>;
        hipError_t ae = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pair_lds);
        if (ae != hipSuccess) return done(fail(GARLIC_ERR_HIP, "LD counts: %s", hipGetErrorString(ae)));
    }
    if (pair_tiled) {   // all chromosomes in one grid
        std::vector<LdPairChr> pc;
        int64_t blocks = 0;
        for (int c = 0; c < p->nchr; c++) {
            pc.push_back(LdPairChr{p->chr_off[c], p->chr_off[c + 1], blocks});
            blocks += (p->chr_nloci[c] + LD_PAIR_T - 1) / LD_PAIR_T;
        }
        DevBuf<LdPairChr> &d_pc = p->lds.pair_chrs;
        if ((rc = d_pc.reserve(pc.size()))) return done(rc);
        e = hipMemcpyAsync(d_pc.p, pc.data(), sizeof(LdPairChr) * pc.size(), hipMemcpyHostToDevice, s);
        if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "LD counts: %s", hipGetErrorString(e)));
        if (phased)
            hipLaunchKernelGGL(ld_pair_tiled_kernel<true>, dim3((unsigned)blocks), dim3(pair_threads), pair_lds, s, d_m.p,
                               d_h.p, d_o.p, p->d_phase.p, nblk, p->nloci, d_pc.p, p->nchr, winsize, pair);
        else
            hipLaunchKernelGGL(ld_pair_tiled_kernel<false>, dim3((unsigned)blocks), dim3(pair_threads), pair_lds, s, d_m.p,
                               d_h.p, (const uint64_t *)nullptr, (const uint64_t *)nullptr, nblk, p->nloci, d_pc.p,
                               p->nchr, winsize, pair);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(s);   // pc (host) is read by the copy above
        if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "LD counts: %s", hipGetErrorString(e)));
    }
    if (pair_mfma) {   // all chromosomes in one grid, 256 SNPs i per workgroup
        std::vector<LdPairChr> pc;
        int64_t blocks = 0;
        for (int c = 0; c < p->nchr; c++) {
            pc.push_back(LdPairChr{p->chr_off[c], p->chr_off[c + 1], blocks});
            blocks += (p->chr_nloci[c] + LDM_TI - 1) / LDM_TI;
        }
        DevBuf<LdPairChr> &d_pc = p->lds.pair_chrs;
        if ((rc = d_pc.reserve(pc.size()))) return done(rc);
        e = hipMemcpyAsync(d_pc.p, pc.data(), sizeof(LdPairChr) * pc.size(), hipMemcpyHostToDevice, s);
        const void *fn = fuse_hr2 ? (mfma_nj <= 2 ? (const void *)ld_pair_mfma_kernel<2, true> : mfma_nj == 3 ? (const void *)ld_pair_mfma_kernel<3, true>
                                     : mfma_nj == 4 ? (const void *)ld_pair_mfma_kernel<4, true> : (const void *)ld_pair_mfma_kernel<5, true>)
                                  : (mfma_nj <= 2 ? (const void *)ld_pair_mfma_kernel<2, false> : mfma_nj == 3 ? (const void *)ld_pair_mfma_kernel<3, false>
                                     : mfma_nj == 4 ? (const void *)ld_pair_mfma_kernel<4, false> : (const void *)ld_pair_mfma_kernel<5, false>);
        const int nj = std::max(2, mfma_nj);
        const size_t lds = std::max((size_t)2 * 2 * (4 + nj - 1) * WAVE * 16, fuse_hr2 ? LDM_XT_BYTES : (size_t)0);
        const double *a_hf = nullptr;
        double *a_c = nullptr;
        if (fuse_hr2) {      // homFreq from the locus counts, room for the combined table (+ 1 KB: ld_sum_col_kernel's last request)
            const size_t n = (size_t)p->nloci * winsize;
            if ((rc = p->lds.hf.reserve(p->nloci)) || (rc = p->lds.fwd.reserve(2 * n + 256))) return done(rc);
            hipLaunchKernelGGL(ld_homfreq_kernel, dim3((unsigned)((p->nloci + 255) / 256)), dim3(256), 0, s, p->lds.loc_planes.p, p->nloci,
                               p->lds.hf.p);
            a_hf = p->lds.hf.p;
            a_c = p->lds.fwd.p;
        }
        if (e == hipSuccess && lds > 48 * 1024) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "LD counts: %s", hipGetErrorString(e)));
        const uint64_t *a_m = d_m.p, *a_h = d_h.p;
        const LdPairChr *a_pc = d_pc.p;
        int a_nblk = nblk, a_nchr = p->nchr, a_w = winsize;
        int64_t a_nloci = p->nloci;
        void *kargs[] = {(void *)&a_m, (void *)&a_h, (void *)&a_nblk, (void *)&a_nloci, (void *)&a_pc, (void *)&a_nchr, (void *)&a_w, (void *)&pair,
                         (void *)&a_hf, (void *)&a_c};
        e = hipLaunchKernel(fn, dim3((unsigned)blocks), dim3(256), kargs, lds, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);   // pc (host) is read by the copy above
        if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "LD counts: %s", hipGetErrorString(e)));
        p->lds.fused_done = fuse_hr2;
    }
    if (pair_lane) {   // all chromosomes in one grid, tiles of 256 SNPs
        std::vector<LdPairChr> pc;
        int64_t blocks = 0;
        for (int c = 0; c < p->nchr; c++) {
            pc.push_back(LdPairChr{p->chr_off[c], p->chr_off[c + 1], blocks});
            blocks += (p->chr_nloci[c] + LD_LANE_T - 1) / LD_LANE_T;
        }
        DevBuf<LdPairChr> &d_pc = p->lds.pair_chrs;
        if ((rc = d_pc.reserve(pc.size()))) return done(rc);
        e = hipMemcpyAsync(d_pc.p, pc.data(), sizeof(LdPairChr) * pc.size(), hipMemcpyHostToDevice, s);
        const void *fn = lane_dc == 16 ? (phased ? (const void *)ld_pair_lane_kernel<true, 16> : (const void *)ld_pair_lane_kernel<false, 16>)
                                       : (phased ? (const void *)ld_pair_lane_kernel<true, 32> : (const void *)ld_pair_lane_kernel<false, 32>);
        if (e == hipSuccess && lane_lds > 48 * 1024) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lane_lds);
        if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "LD counts: %s", hipGetErrorString(e)));
        const uint64_t *a_m = d_m.p, *a_h = d_h.p, *a_o = phased ? d_o.p : nullptr, *a_f = phased ? p->d_phase.p : nullptr;
        const LdPairChr *a_pc = d_pc.p;
        int a_nblk = nblk, a_nchr = p->nchr, a_w = winsize, a_stage = lane_stage;
        int64_t a_nloci = p->nloci;
        void *kargs[] = {(void *)&a_m, (void *)&a_h, (void *)&a_o, (void *)&a_f, (void *)&a_nblk, (void *)&a_nloci, (void *)&a_pc,
                         (void *)&a_nchr, (void *)&a_w, (void *)&a_stage, (void *)&pair};
        e = hipLaunchKernel(fn, dim3((unsigned)blocks), dim3(LD_LANE_T), kargs, lane_lds, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);   // pc (host) is read by the copy above
        if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "LD counts: %s", hipGetErrorString(e)));
    }
    if (pair_flat) {
        const unsigned grid = (unsigned)(((int64_t)p->nloci * winsize + 255) / 256);
        if (phased)
            hipLaunchKernelGGL(ld_pair_flat_kernel<true>, dim3(grid), dim3(256), 0, s, d_m.p, d_h.p, d_o.p, p->d_phase.p, nblk,
                               p->nloci, p->d_chr_off.p, p->nchr, winsize, pair);
        else
            hipLaunchKernelGGL(ld_pair_flat_kernel<false>, dim3(grid), dim3(256), 0, s, d_m.p, d_h.p, (const uint64_t *)nullptr,
                               (const uint64_t *)nullptr, nblk, p->nloci, p->d_chr_off.p, p->nchr, winsize, pair);
    }
    for (int c = 0; !pair_tiled && !pair_flat && !pair_lane && !pair_mfma && c < p->nchr; c++) {
        if (phased)
            hipLaunchKernelGGL(ld_pair_phased_kernel, dim3((unsigned)p->chr_nloci[c]), dim3(256), 0, s, d_m.p,
                               d_h.p, d_o.p, p->d_phase.p, nblk, p->nloci, p->chr_off[c], p->chr_off[c + 1],
                               winsize, pair);
        else
            hipLaunchKernelGGL(ld_pair_kernel, dim3((unsigned)p->chr_nloci[c]), dim3(256), 0, s, d_m.p, d_h.p,
                               nblk, p->nloci, p->chr_off[c], p->chr_off[c + 1], winsize, pair);
    }
    e = hipGetLastError();
    if (e == hipSuccess && where == GARLIC_HOST) {
        e = hipMemcpyAsync(locus_counts, loc, sizeof(int32_t) * p->nloci * 2, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess)
            e = hipMemcpyAsync(pair_counts, pair, sizeof(int32_t) * npair, hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "LD counts: %s", hipGetErrorString(e)));
    return done(GARLIC_OK);
}

int garlic_ld_finish(garlic_panel *p, int32_t winsize, int32_t phased, const int32_t *locus_counts,
                     const int32_t *pair_counts, double *ld_out, int32_t where)
{
    int rc;
    if ((rc = ld_check(p, winsize, phased))) return rc;
    if (!locus_counts || !pair_counts) return fail(GARLIC_ERR_INVALID, "count buffers are required");
    hipStream_t s = p->ctx->stream;
    const size_t n = (size_t)p->nloci * winsize;
    DevBuf<int32_t> &d_loc = p->lds.loc, &d_pair = p->lds.pair;
    DevBuf<double> &d_hf = p->lds.hf, &d_fwd = p->lds.fwd, &d_bwd = p->lds.bwd, &d_ld = p->lds.ld;
    DevBuf<LdSumChr> &d_sum_chrs = p->lds.sum_chrs;
    auto done = [&](int code) { return code; };   // the scratch stays with the panel
    const int32_t *loc = locus_counts, *pair = pair_counts;
    hipError_t e = hipSuccess;
    if (where == GARLIC_HOST) {
        if ((rc = d_loc.reserve((size_t)p->nloci * 2)) || (rc = d_pair.reserve(n * 2))) return done(rc);
        e = hipMemcpyAsync(d_loc.p, locus_counts, sizeof(int32_t) * p->nloci * 2, hipMemcpyHostToDevice, s);
        if (e == hipSuccess)
            e = hipMemcpyAsync(d_pair.p, pair_counts, sizeof(int32_t) * n * 2, hipMemcpyHostToDevice, s);
        loc = d_loc.p; pair = d_pair.p;
    }
    // narrow windows (GARLIC's default --winsize is 10): hr2 evaluated in place, one thread per (window start, column)
    if (winsize <= LD_SMALL_MAX_W && !getenv("GARLIC_LD_NO_FLAT")) {
        if ((rc = d_hf.reserve(p->nloci))) return done(rc);
        double *ld = ld_out;
        if (where == GARLIC_HOST || !ld_out) {
            if ((rc = d_ld.reserve(n))) return done(rc);
            ld = d_ld.p;
        }
        if (e == hipSuccess) e = hipMemsetAsync(ld, 0, sizeof(double) * n, s);      // initLDData zero-fills
        if (e == hipSuccess && phased) e = hipMemcpyAsync(d_hf.p, p->freq.data(), sizeof(double) * p->nloci, hipMemcpyHostToDevice, s);
        if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "LD finish: %s", hipGetErrorString(e)));
        if (!phased)
            hipLaunchKernelGGL(ld_homfreq_kernel, dim3((unsigned)((p->nloci + 255) / 256)), dim3(256), 0, s, loc, p->nloci, d_hf.p);
        garlic_ctx *ctx = p->ctx;
        const int slot = (int)(ctx->n_calls % garlic_ctx::HIST);
        (void)hipEventRecord(ctx->hist0[slot], s);
        hipLaunchKernelGGL(ld_sum_flat_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pair, d_hf.p, p->d_chr_off.p,
                           p->nchr, p->nloci, winsize, ld);
        (void)hipEventRecord(ctx->hist1[slot], s);
        ctx->n_calls++;
        e = hipGetLastError();
        if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "LD finish: %s", hipGetErrorString(e)));
        if ((rc = install_ld(p, winsize, ld, false))) return done(rc);
        if (where == GARLIC_HOST && ld_out) {
            e = hipMemcpyAsync(ld_out, ld, sizeof(double) * n, hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "LD finish: %s", hipGetErrorString(e)));
        }
        return done(GARLIC_OK);
    }
    // ordered sums: LDS-tiled kernel (one thread per column of the LD row) unless the window is too wide
    // ... thread = SNP of the window, accumulators = window starts (ld_sum_col_kernel: 32 < W <= 512) unless switched off
    const int col_threads = (winsize + 16 + WAVE - 1) / WAVE * WAVE;
    const bool by_snp = ld_sums_by_snp(winsize);
    const bool have_table = p->lds.fused_done && by_snp;      // the pair kernel has written the hr2 table (garlic_panel_compute_ld)
    p->lds.fused_done = false;
    const bool tiled = by_snp || (winsize <= LD_SUM_MAX_W && !getenv("GARLIC_LD_SUM_L2"));
    const int sum_b = by_snp ? std::min(LD_COL_B, col_threads - winsize) : LD_SUM_B;      // (thread W + B - 1 reads one element further on odd steps)
    // (the SNP-per-thread kernel reads one combined row of 2W doubles per SNP, in d_fwd; + 1 KB the last row's
    // last request may run over)
    if ((rc = d_hf.reserve(p->nloci)) || (rc = d_fwd.reserve(by_snp ? 2 * n + 256 : n)) || (!by_snp && (rc = d_bwd.reserve(n))))
        return done(rc);
    double *ld = ld_out;
    // nobody asked for the LD matrix itself and the sum kernel writes the wLOD weights directly: it is not made at all
    const bool weights_only = by_snp && !ld_out;
    if (!weights_only && (where == GARLIC_HOST || !ld_out)) {
        if ((rc = d_ld.reserve(n))) return done(rc);
        ld = d_ld.p;
    }
    // initLDData zero-fills; ld_sum_col_kernel writes every entry of the window starts that have a full window, which
    // leaves the last W - 1 rows of each chromosome
    if (weights_only) {
    } else if (by_snp) {
        for (int c = 0; c < p->nchr && e == hipSuccess; c++) {
            const int64_t lo = p->chr_off[c], hi = p->chr_off[c + 1], from = std::max(lo, hi - winsize + 1);
            if (hi > from) e = hipMemsetAsync(ld + from * winsize, 0, sizeof(double) * (size_t)(hi - from) * winsize, s);
        }
    } else {
        e = hipMemsetAsync(ld, 0, sizeof(double) * n, s);
    }
    if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "LD finish: %s", hipGetErrorString(e)));
    if (phased) {       // r2 takes FreqData::freq where hr2 takes homFreq (garlic-data.cpp:587-588)
        e = hipMemcpyAsync(d_hf.p, p->freq.data(), sizeof(double) * p->nloci, hipMemcpyHostToDevice, s);
        if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "LD finish: %s", hipGetErrorString(e)));
    } else {
        hipLaunchKernelGGL(ld_homfreq_kernel, dim3((unsigned)((p->nloci + 255) / 256)), dim3(256), 0, s, loc,
                           p->nloci, d_hf.p);
    }
    std::vector<LdSumChr> sum_chrs;
    int64_t sum_blocks = 0;
    for (int c = 0; tiled && c < p->nchr; c++) {
        const int64_t nstarts = p->chr_off[c + 1] - p->chr_off[c] - winsize + 1;
        if (nstarts < 1) continue;
        sum_chrs.push_back(LdSumChr{p->chr_off[c], nstarts, sum_blocks});
        sum_blocks += (nstarts + sum_b - 1) / sum_b;
    }
    if (tiled && (rc = d_sum_chrs.reserve(std::max<size_t>(sum_chrs.size(), 1)))) return done(rc);
    if (by_snp && (rc = reserve_skew(p, winsize, false))) return done(rc);      // the sum kernel writes the wLOD weights as well
    for (int c = 0; c < p->nchr; c++) {
        const int64_t lo = p->chr_off[c], hi = p->chr_off[c + 1];
        const size_t hr2_lds = sizeof(double) * (LD_HR2_T + winsize + (size_t)LD_HR2_T * (winsize + 1));
        if (have_table)
            ;
        else if (by_snp && hr2_lds <= 64 * 1024 && !getenv("GARLIC_LD_HR2_PLAIN"))
            hipLaunchKernelGGL(ld_hr2_tile_kernel, dim3((unsigned)((hi - lo + LD_HR2_T - 1) / LD_HR2_T)), dim3(256),
                               hr2_lds, s, pair, d_hf.p, lo, hi, winsize, d_fwd.p);
        else if (by_snp)
            hipLaunchKernelGGL(ld_hr2_kernel<true>, dim3((unsigned)(hi - lo)), dim3(128), 0, s, pair, d_hf.p, lo, hi,
                               winsize, d_fwd.p, (double *)nullptr);
        else
            hipLaunchKernelGGL(ld_hr2_kernel<false>, dim3((unsigned)(hi - lo)), dim3(256), 0, s, pair, d_hf.p, lo, hi,
                               winsize, d_fwd.p, d_bwd.p);
        if (hi - lo >= winsize && !tiled)
            hipLaunchKernelGGL(ld_sum_kernel, dim3((unsigned)(hi - lo - winsize + 1)), dim3(256), 0, s, d_fwd.p, d_bwd.p,
                               lo, winsize, ld);
    }
    if (tiled && !sum_chrs.empty()) {   // all chromosomes in one grid, after every hr2 value exists
        e = hipMemcpyAsync(d_sum_chrs.p, sum_chrs.data(), sizeof(LdSumChr) * sum_chrs.size(), hipMemcpyHostToDevice, s);
        if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "LD finish: %s", hipGetErrorString(e)));
        const int threads = by_snp ? col_threads : (winsize + WAVE - 1) / WAVE * WAVE;
        const int col_pieces = (threads * 8 + 1023) / 1024;
        const size_t lds = by_snp ? sizeof(double) * std::max<size_t>((size_t)LD_COL_BATCH * LD_COL_NBATCH * col_pieces * 128 + 130, (size_t)threads * 17)
                                  : sizeof(double) * 2 * (2 * (size_t)winsize - 1 + 128 + threads);
        // (the call's dominant kernel: its HIP-event time is what garlic_recent_kernel_ms reports for an LD call)
        garlic_ctx *ctx = p->ctx;
        const int slot = (int)(ctx->n_calls % garlic_ctx::HIST);
        (void)hipEventRecord(ctx->hist0[slot], s);
        if (by_snp) {
            static_assert(LD_COL_MAX_THREADS <= 128 * LD_COL_MAX_PIECES, "one instantiation per request count");
            const void *fn = col_pieces == 1 ? (const void *)ld_sum_col_kernel<1> : col_pieces == 2 ? (const void *)ld_sum_col_kernel<2>
                           : col_pieces == 3 ? (const void *)ld_sum_col_kernel<3> : col_pieces == 4 ? (const void *)ld_sum_col_kernel<4>
                                                                                                    : (const void *)ld_sum_col_kernel<5>;
            if (lds > 48 * 1024) {
                e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "LD finish: %s", hipGetErrorString(e)));
            }
            const double *a_c = d_fwd.p;
            const LdSumChr *a_chrs = d_sum_chrs.p;
            int a_nchr = (int)sum_chrs.size(), a_w = winsize, a_b = sum_b;
            double *a_ld = ld, *a_d = p->d_skew.p + SKEW_FRONT;
            unsigned a_nwork = (unsigned)sum_blocks;
            void *kargs[] = {(void *)&a_c, (void *)&a_chrs, (void *)&a_nchr, (void *)&a_w, (void *)&a_b, (void *)&a_ld, (void *)&a_d, (void *)&a_nwork};
            e = hipLaunchKernel(fn, dim3((a_nwork + 7u) / 8u * 8u), dim3(threads), kargs, lds, s);
            if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "LD finish: %s", hipGetErrorString(e)));
        }
        else
            hipLaunchKernelGGL(ld_sum_tiled_kernel, dim3((unsigned)sum_blocks), dim3(threads), lds, s, d_fwd.p, d_bwd.p,
                               d_sum_chrs.p, (int)sum_chrs.size(), winsize, ld);
        (void)hipEventRecord(ctx->hist1[slot], s);
        ctx->n_calls++;
    }
    e = hipGetLastError();
    if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "LD finish: %s", hipGetErrorString(e)));
    if ((rc = install_ld(p, winsize, ld, by_snp))) return done(rc);
    if (where == GARLIC_HOST && ld_out) {
        e = hipMemcpyAsync(ld_out, ld, sizeof(double) * n, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "LD finish: %s", hipGetErrorString(e)));
    }
    return done(GARLIC_OK);
}

int garlic_panel_compute_ld(garlic_panel *p, int32_t winsize, int32_t phased, const int32_t *sub_idx,
                            int32_t n_sub, double *ld_out, int32_t where)
{
    int rc;
    if ((rc = ld_check(p, winsize, phased))) return rc;
    DevBuf<int32_t> &d_loc = p->lds.loc, &d_pair = p->lds.pair;   // kept with the panel, as all LD scratch
    // every individual's counts are here: the pair kernel can go on to the hr2 values, no pair table (GARLIC_LD_UNFUSED:
    // the two steps of garlic_ld_counts / garlic_ld_finish, which a sharded panel needs)
    const bool fuse = ld_pairs_on_mfma(winsize, phased) && ld_sums_by_snp(winsize) && !getenv("GARLIC_LD_UNFUSED");
    if ((rc = d_loc.reserve((size_t)p->nloci * 2)) || (rc = d_pair.reserve(fuse ? 2 : (size_t)p->nloci * winsize * 2)))
        return rc;
    p->lds.fuse_request = fuse;
    if ((rc = garlic_ld_counts(p, winsize, phased, sub_idx, n_sub, d_loc.p, d_pair.p, GARLIC_DEVICE))) {
        p->lds.fuse_request = false;
        return rc;
    }
    if (where == GARLIC_DEVICE || !ld_out)
        return garlic_ld_finish(p, winsize, phased, d_loc.p, d_pair.p, ld_out, GARLIC_DEVICE);
    // host output: finish on the device, then copy out
    DevBuf<double> &d_ld = p->lds.ld;
    if ((rc = d_ld.reserve((size_t)p->nloci * winsize))) return rc;
    rc = garlic_ld_finish(p, winsize, phased, d_loc.p, d_pair.p, d_ld.p, GARLIC_DEVICE);
    if (rc == GARLIC_OK) {
        hipError_t e = hipMemcpy(ld_out, d_ld.p, sizeof(double) * p->nloci * winsize, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(GARLIC_ERR_HIP, "LD copy-out: %s", hipGetErrorString(e));
    }
    return rc;
}

// Drops everything the panel keeps only to make the next call cheaper: LD scratch, the score and
// feed scratch of host-output / feed calls, staging buffers.  Inputs, tables, LD weights and the
// TGLS term matrix stay.
int garlic_panel_release_scratch(garlic_panel *p)
{
    if (!p) return fail(GARLIC_ERR_INVALID, "panel is NULL");
    int rc;
    if ((rc = set_device(p->ctx))) return rc;
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    p->lds.release();
    p->d_out.release(); p->d_feed.release();
    p->d_stage16.release(); p->d_stage64.release();
    return GARLIC_OK;
}

int garlic_lod_out_layout(garlic_panel *p, int32_t pitch_align, int32_t nind_out, int64_t *chr_base,
                          int64_t *chr_pitch, int64_t *total)
{
    if (!p) return fail(GARLIC_ERR_INVALID, "panel is NULL");
    if (pitch_align < 1 || nind_out < 1)
        return fail(GARLIC_ERR_INVALID, "pitch_align and nind_out must be >= 1");
    Layout L = make_layout(p, pitch_align, nind_out);
    for (int c = 0; c < p->nchr; c++) {
        if (chr_base) chr_base[c] = L.base[c];
        if (chr_pitch) chr_pitch[c] = L.pitch[c];
    }
    if (total) *total = L.total;
    return GARLIC_OK;
}

int garlic_lod_windows(garlic_panel *p, int32_t winsize, double error, int32_t max_gap,
                       int32_t use_gl, int32_t ind_begin, int32_t ind_count, int32_t pitch_align,
                       double *out, int32_t where)
{
    if (!p) return fail(GARLIC_ERR_INVALID, "panel is NULL");
    return launch_lod(p, use_gl ? MODE_LOD_GL : MODE_LOD, winsize, error, max_gap, 0, 0.0, ind_begin,
                      ind_count, pitch_align, out, where);
}

int garlic_lod_windows_multi(garlic_panel *p, const int32_t *winsizes, int32_t n_winsizes, double error,
                             int32_t max_gap, int32_t use_gl, int32_t ind_begin, int32_t ind_count,
                             int32_t pitch_align, double *out, int64_t out_stride, int32_t where)
{
    if (!p || !winsizes) return fail(GARLIC_ERR_INVALID, "panel and winsizes are required");
    if (n_winsizes < 1) return fail(GARLIC_ERR_INVALID, "n_winsizes must be >= 1");
    if (!out) return fail(GARLIC_ERR_INVALID, "out is NULL");
    const Layout L = make_layout(p, pitch_align, ind_count);
    if (out_stride < L.total)
        return fail(GARLIC_ERR_INVALID, "out_stride %lld smaller than one window size's scores (%lld doubles)",
                    (long long)out_stride, (long long)L.total);
    for (int32_t k = 0; k < n_winsizes; k++) {   // the panel stays resident: only the work list changes
        const int rc = launch_lod(p, use_gl ? MODE_LOD_GL : MODE_LOD, winsizes[k], error, max_gap, 0, 0.0, ind_begin,
                                  ind_count, pitch_align, out + (int64_t)k * out_stride, where);
        if (rc) return rc;
    }
    return GARLIC_OK;
}

int garlic_wlod_windows(garlic_panel *p, int32_t winsize, double error, int32_t max_gap, int32_t use_gl,
                        int32_t M, double mu, int32_t ind_begin, int32_t ind_count, int32_t pitch_align,
                        double *out, int32_t where)
{
    if (!p) return fail(GARLIC_ERR_INVALID, "panel is NULL");
    p->wlod_use_gl = use_gl != 0;
    return launch_lod(p, MODE_WLOD, winsize, error, max_gap, M, mu, ind_begin, ind_count, pitch_align,
                      out, where);
}

// thinned > 0: `scores` is already the thinned matrix of make_layout(.., thinned) (every column is a
// sample); else the full score matrix, sampled every `step` loci
static int flatten_impl(garlic_panel *p, const double *scores, int32_t pitch_align, int32_t nind_out,
                        int32_t step, double *feed, int64_t feed_capacity, int64_t *count, int64_t *chr_counts,
                        int32_t thinned = 0, const int32_t *d_ind_list = nullptr, int32_t n_list = 0)
{   // d_ind_list (device): the rows of the feed, in this order inside every chromosome; NULL: 0 .. nind_out-1
    if (!p || !scores || !count) return fail(GARLIC_ERR_INVALID, "panel, scores and count are required");
    if (step < 1 || pitch_align < 1 || nind_out < 1)
        return fail(GARLIC_ERR_INVALID, "step, pitch_align and nind_out must be >= 1");
    int rc;
    if ((rc = set_device(p->ctx))) return rc;
    hipStream_t s = p->ctx->stream;
    Layout L = make_layout(p, pitch_align, nind_out, thinned);
    std::vector<ChrDev> chrs(p->nchr);
    for (int c = 0; c < p->nchr; c++) {
        const int32_t cols = thinned > 0 ? (p->chr_nloci[c] + thinned - 1) / thinned : p->chr_nloci[c];
        chrs[c] = ChrDev{p->chr_off[c], L.base[c], L.pitch[c], cols, 0};
    }
    if (thinned > 0) step = 1;
    const int per_chr = d_ind_list ? n_list : nind_out;
    const int nrows = p->nchr * per_chr;
    if ((rc = p->d_chrs.reserve(chrs.size()))) return rc;
    if ((rc = p->d_row_counts.reserve((size_t)nrows))) return rc;
    p->plan.valid = false; // d_chrs is shared with the work plan
    HIP_TRY(hipMemcpyAsync(p->d_chrs.p, chrs.data(), sizeof(ChrDev) * chrs.size(), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(feed_count_kernel, dim3((unsigned)nrows), dim3(WAVE), 0, s, scores, p->d_chrs.p, p->nchr,
                       per_chr, d_ind_list, step, p->d_row_counts.p);
    std::vector<int64_t> counts((size_t)nrows);
    HIP_TRY(hipMemcpyAsync(counts.data(), p->d_row_counts.p, sizeof(int64_t) * nrows, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    int64_t total = 0; // exclusive scan over (chromosome, individual) rows: tiny, done on the host
    if (chr_counts)
        for (int c = 0; c < p->nchr; c++) {
            chr_counts[c] = 0;
            for (int i = 0; i < per_chr; i++) chr_counts[c] += counts[(size_t)c * per_chr + i];
        }
    for (auto &c : counts) { const int64_t n = c; c = total; total += n; }
    *count = total;
    if (total > feed_capacity || total == 0) return GARLIC_OK;
    if (!feed) return fail(GARLIC_ERR_INVALID, "feed is NULL");
    HIP_TRY(hipMemcpyAsync(p->d_row_counts.p, counts.data(), sizeof(int64_t) * nrows, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(feed_write_kernel, dim3((unsigned)nrows), dim3(WAVE), 0, s, scores, p->d_chrs.p, p->nchr,
                       per_chr, d_ind_list, step, p->d_row_counts.p, feed);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    return GARLIC_OK;
}

int garlic_lod_flatten(garlic_panel *p, const double *scores, int32_t pitch_align, int32_t nind_out,
                       int32_t step, double *feed, int64_t feed_capacity, int64_t *count)
{
    return flatten_impl(p, scores, pitch_align, nind_out, step, feed, feed_capacity, count, nullptr);
}

// LOD / wLOD scores and their thinned KDE feed in one call: the scores never leave the device.  The general form:
// the scores (full, or the chain kernel's thinned matrix) into device scratch, then garlic_lod_flatten's two passes.
static int feed_single(garlic_panel *p, int32_t winsize, double error, int32_t max_gap, int32_t use_gl,
                       int32_t weighted, int32_t M, double mu, int32_t step, const int32_t *ind_idx, int32_t n_idx,
                       double *feed, int64_t feed_capacity, int64_t *count, int64_t *chr_counts)
{
    if (!p || !count) return fail(GARLIC_ERR_INVALID, "panel and count are required");
    if (step < 1) return fail(GARLIC_ERR_INVALID, "step must be >= 1");
    if (ind_idx && n_idx < 1) return fail(GARLIC_ERR_INVALID, "an individual list needs at least one entry");
    int rc;
    if ((rc = set_device(p->ctx))) return rc;
    // the listed individuals (convertSubsetWinData2DoubleData's randInd[]) and the blocks that hold them
    const int nblk = (p->nind + WAVE - 1) / WAVE;
    std::vector<uint8_t> blocks;
    DevBuf<int32_t> d_list;
    auto done = [&](int code) { d_list.release(); return code; };
    if (ind_idx) {
        blocks.assign((size_t)nblk, 0);
        std::vector<uint8_t> seen((size_t)p->nind, 0);
        for (int k = 0; k < n_idx; k++) {
            const int i = ind_idx[k];
            if (i < 0 || i >= p->nind) return fail(GARLIC_ERR_INVALID, "feed individual %d outside panel of %d", i, p->nind);
            if (seen[(size_t)i]) return fail(GARLIC_ERR_INVALID, "feed individual %d listed twice", i);
            seen[(size_t)i] = 1;
            blocks[(size_t)(i >> 6)] = 1;
        }
        if ((rc = d_list.reserve((size_t)n_idx))) return done(rc);
        hipError_t e = hipMemcpy(d_list.p, ind_idx, sizeof(int32_t) * (size_t)n_idx, hipMemcpyHostToDevice);
        if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "feed: %s", hipGetErrorString(e)));
    }
    const int32_t n_rows = ind_idx ? n_idx : p->nind;
    // Unweighted --error scores with a real thinning step: the chain kernel stores only the sampled
    // windows (8/step B per window instead of 8 B, no full-size scratch).  Otherwise the full scores
    // go to the panel's device scratch (the one host-output calls use; it stays allocated, hipMalloc
    // of 8 GB per call would cost more than the kernels) and are sampled from there.
    int32_t thinned = (!weighted && !use_gl && step >= 4) ? step : 0;
    if (thinned) {   // the exact chain (lod_exact_needed) writes full scores only
        if (!p->have_freq) return done(fail(GARLIC_ERR_STATE, "panel needs map, freq and genotypes before computing LOD"));
        if ((rc = ensure_term_table(p, error))) return done(rc);
        if (lod_exact_needed(p, MODE_LOD, winsize)) thinned = 0;
    }
    const Layout L = make_layout(p, 32, p->nind, thinned);
    garlic_panel::ScoreBuf &scores = p->d_out;
    DevBuf<double> &d_feed = p->d_feed;            // kept with the panel: window-size sweeps call this repeatedly
    if ((rc = scores.reserve(p->ctx, (size_t)L.total))) return done(rc);
    if (weighted) p->wlod_use_gl = use_gl != 0;
    rc = launch_lod(p, weighted ? MODE_WLOD : (use_gl ? MODE_LOD_GL : MODE_LOD), winsize, error, max_gap, M, mu, 0,
                    p->nind, 32, scores.p, GARLIC_DEVICE, thinned, ind_idx ? &blocks : nullptr);
    if (rc) return done(rc);
    // at most ceil(nloci_c / step) values per (chromosome, individual)
    int64_t cap = 0;
    for (int c = 0; c < p->nchr; c++) cap += ((int64_t)p->chr_nloci[c] + step - 1) / step * n_rows;
    if ((rc = d_feed.reserve((size_t)std::max<int64_t>(cap, 1)))) return done(rc);
    if ((rc = flatten_impl(p, scores.p, 32, p->nind, step, d_feed.p, cap, count, chr_counts, thinned,
                           ind_idx ? d_list.p : nullptr, n_idx)))
        return done(rc);
    if (*count > feed_capacity || *count == 0) return done(GARLIC_OK);
    if (!feed) return done(fail(GARLIC_ERR_INVALID, "feed is NULL"));
    hipError_t e = hipMemcpy(feed, d_feed.p, sizeof(double) * (size_t)*count, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "feed copy-out: %s", hipGetErrorString(e)));
    return done(GARLIC_OK);
}

int garlic_lod_feed_subset(garlic_panel *p, int32_t winsize, double error, int32_t max_gap, int32_t use_gl,
                           int32_t weighted, int32_t M, double mu, int32_t step, const int32_t *ind_idx, int32_t n_idx,
                           double *feed, int64_t feed_capacity, int64_t *count, int64_t *chr_counts)
{
    if (!p || !count) return fail(GARLIC_ERR_INVALID, "panel and count are required");
    // unweighted --error scores with a real thinning step: the chain kernel writes the feed itself (garlic_lod_feed_multi)
    if (!weighted && !use_gl && step >= 4)
        return garlic_lod_feed_multi(p, &winsize, &step, 1, error, max_gap, ind_idx, n_idx, &feed, &feed_capacity, count, chr_counts);
    return feed_single(p, winsize, error, max_gap, use_gl, weighted, M, mu, step, ind_idx, n_idx, feed, feed_capacity, count,
                       chr_counts);
}

int garlic_lod_feed(garlic_panel *p, int32_t winsize, double error, int32_t max_gap, int32_t use_gl,
                    int32_t weighted, int32_t M, double mu, int32_t step, double *feed,
                    int64_t feed_capacity, int64_t *count, int64_t *chr_counts)
{
    return garlic_lod_feed_subset(p, winsize, error, max_gap, use_gl, weighted, M, mu, step, nullptr, 0, feed,
                                  feed_capacity, count, chr_counts);
}

// The KDE feeds of several window sizes in one call (exploreWinsizes / selectWinsizeFromList run the same panel
// through every size of --winsize-multi, src/garlic-roh.cpp:726-751, 881-920).
//
// The window mask depends on positions only (the same for every individual), so which sampled loci hold a score --
// and the place of every sample in convertWinData2DoubleData's chromosome -> individual -> locus order -- is known
// before anything is computed: lod_feed_kernel stores every sample straight into the feed (row = position in the
// individual list, column = rank among the chromosome's scored samples).  No thinned score matrix, no compaction
// pass.  (That needs every scored window to be a finite number other than -9999, i.e. finite terms and
// !lod_exact_needed; otherwise, and for steps below 4, the single-size path runs: scores, then garlic_lod_flatten.)
// Each size has a stream and a feed buffer of its own and every size's kernel is enqueued before the first feed is
// fetched: the tail of one size's chain kernel (its longest runs, a few waves per CU) runs beside the bulk of the
// next size's, and a feed crosses PCIe while the following sizes are computed.
int garlic_lod_feed_multi(garlic_panel *p, const int32_t *winsizes, const int32_t *steps, int32_t n_sizes, double error,
                          int32_t max_gap, const int32_t *ind_idx, int32_t n_idx, double *const *feeds,
                          const int64_t *feed_capacity, int64_t *counts, int64_t *chr_counts)
{
    if (!p || !winsizes || !steps || !feeds || !feed_capacity || !counts)
        return fail(GARLIC_ERR_INVALID, "panel, winsizes, steps, feeds, feed_capacity and counts are required");
    if (n_sizes < 1) return fail(GARLIC_ERR_INVALID, "n_sizes must be >= 1");
    if (ind_idx && n_idx < 1) return fail(GARLIC_ERR_INVALID, "an individual list needs at least one entry");
    for (int i = 0; i < n_sizes; i++) {
        if (winsizes[i] <= 1) return fail(GARLIC_ERR_INVALID, "SNP window size must be > 1 (got %d)", winsizes[i]);
        if (steps[i] < 1) return fail(GARLIC_ERR_INVALID, "step must be >= 1");
    }
    garlic_ctx *ctx = p->ctx;
    int rc;
    if ((rc = set_device(ctx))) return rc;
    if (!p->have_map || !p->have_freq || !p->have_geno)
        return fail(GARLIC_ERR_STATE, "panel needs map, freq and genotypes before computing LOD");
    bool direct = !getenv("GARLIC_FEED_SERIAL");
    for (int i = 0; i < n_sizes && direct; i++) direct = steps[i] >= 4;
    if (direct) {
        if ((rc = ensure_segments(p, max_gap))) return rc;
        if ((rc = ensure_term_table(p, error))) return rc;
        direct = p->tab_all_finite;
        for (int i = 0; i < n_sizes && direct; i++) direct = !lod_exact_needed(p, MODE_LOD, winsizes[i]);
    }
    if (!direct) {
        for (int i = 0; i < n_sizes; i++)
            if ((rc = feed_single(p, winsizes[i], error, max_gap, 0, 0, 0, 0.0, steps[i], ind_idx, n_idx, feeds[i],
                                  feed_capacity[i], &counts[i], chr_counts ? chr_counts + (size_t)i * p->nchr : nullptr)))
                return rc;
        return GARLIC_OK;
    }
    // rows of the feed: the listed individuals in list order, or everyone
    const int nblk = (p->nind + WAVE - 1) / WAVE;
    const int nrows = ind_idx ? n_idx : p->nind;
    std::vector<uint8_t> blocks;
    std::vector<int32_t> row_map;
    DevBuf<int32_t> d_rowmap;
    // one way out: whatever was enqueued on the sizes' streams has finished (the next call reuses their scratch) and the
    // row map is released
    auto done = [&](int code) {
        if (code != GARLIC_OK)
            for (auto *sl : p->feed_slots) (void)hipStreamSynchronize(sl->stream);
        d_rowmap.release();
        return code;
    };
#define FEED_TRY(expr)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return done(fail(GARLIC_ERR_HIP, "feed: %s: %s", #expr, hipGetErrorString(e_))); \
    } while (0)
    if (ind_idx) {
        blocks.assign((size_t)nblk, 0);
        row_map.assign((size_t)p->nind, -1);
        for (int k = 0; k < n_idx; k++) {
            const int i = ind_idx[k];
            if (i < 0 || i >= p->nind) return fail(GARLIC_ERR_INVALID, "feed individual %d outside panel of %d", i, p->nind);
            if (row_map[(size_t)i] >= 0) return fail(GARLIC_ERR_INVALID, "feed individual %d listed twice", i);
            row_map[(size_t)i] = k;
            blocks[(size_t)(i >> 6)] = 1;
        }
        if ((rc = d_rowmap.reserve((size_t)p->nind))) return done(rc);
        hipError_t e = hipMemcpy(d_rowmap.p, row_map.data(), sizeof(int32_t) * (size_t)p->nind, hipMemcpyHostToDevice);
        if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "feed: %s", hipGetErrorString(e)));
    }
    while ((int)p->feed_slots.size() < n_sizes) {
        auto *sl = new garlic_panel::FeedSlot;            // (joins the panel's slots only once it is complete)
        bool ok = hipStreamCreateWithFlags(&sl->stream, hipStreamNonBlocking) == hipSuccess;
        const bool ok0 = ok && hipEventCreate(&sl->ev0) == hipSuccess;
        const bool ok1 = ok0 && hipEventCreate(&sl->ev1) == hipSuccess;
        if (!ok1) {
            if (ok0) (void)hipEventDestroy(sl->ev0);
            if (ok) (void)hipStreamDestroy(sl->stream);
            delete sl;
            return done(fail(GARLIC_ERR_HIP, "feed: stream / event creation failed"));
        }
        p->feed_slots.push_back(sl);
    }
    FEED_TRY(hipStreamSynchronize(ctx->stream));          // uploads and earlier calls on the context's stream
    // ---- plans and their uploads, all sizes, before any kernel is enqueued (an upload from pageable memory waits
    //      for the device to take it: behind a running chain kernel it would hold back the sizes that follow)
    std::vector<size_t> n_items((size_t)n_sizes, 0);
    std::vector<int> grids((size_t)n_sizes, 1);      // workgroups each size's kernel is launched with (feed_grid)
    std::vector<int64_t> total((size_t)n_sizes, 0);
    for (int i = 0; i < n_sizes; i++) {
        garlic_panel::FeedSlot &sl = *p->feed_slots[(size_t)i];
        const int32_t W = winsizes[i], step = steps[i];
        std::vector<Run> runs;
        std::vector<FillItem> fill;
        int64_t n_valid = 0;
        plan_runs(p, W, runs, fill, n_valid);             // in chromosome and position order
        // rank of every run's first sample among its chromosome's scored samples; samples per chromosome
        std::vector<int32_t> col0(runs.size(), -1);
        std::vector<int64_t> nkeep((size_t)p->nchr, 0);
        for (size_t r = 0; r < runs.size(); r++) {
            const int64_t s = ((int64_t)runs[r].a + step - 1) / step * step;
            if (s > runs[r].b) continue;
            col0[r] = (int32_t)nkeep[(size_t)runs[r].chr];
            nkeep[(size_t)runs[r].chr] += (runs[r].b - s) / step + 1;
        }
        std::vector<int> order(runs.size());
        for (size_t k = 0; k < runs.size(); k++) order[k] = (int)k;
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return (runs[x].b - runs[x].a) > (runs[y].b - runs[y].a); });
        std::vector<FeedItem> items;
        build_feed_items(runs, order, ind_idx ? &blocks : nullptr, nblk, col0, items);
        std::vector<ChrDev> chrs((size_t)p->nchr);
        int64_t off = 0;
        for (int c = 0; c < p->nchr; c++) {
            chrs[(size_t)c] = ChrDev{p->chr_off[c], off, nkeep[(size_t)c], p->chr_nloci[c], 1};
            if (chr_counts) chr_counts[(size_t)i * p->nchr + c] = nkeep[(size_t)c] * nrows;
            if (nkeep[(size_t)c] * 8 * (int64_t)nrows >= (int64_t)1 << 32)
                return done(fail(GARLIC_ERR_INVALID, "chromosome %d: feed rows beyond 32-bit offsets", c));
            off += nkeep[(size_t)c] * nrows;
        }
        total[(size_t)i] = off;
        counts[i] = off;
        n_items[(size_t)i] = items.size();
        if (off > feed_capacity[i] || off == 0) { n_items[(size_t)i] = 0; continue; }
        if (!items.empty() && (rc = feed_grid(ctx, items, &grids[(size_t)i], nullptr))) return done(rc);
        if (!feeds[i]) return done(fail(GARLIC_ERR_INVALID, "feed %d is NULL", i));
        if ((rc = sl.items.reserve(std::max<size_t>(items.size(), 1)))) return done(rc);
        if ((rc = sl.chrs.reserve((size_t)p->nchr))) return done(rc);
        if ((rc = sl.counter.reserve(4))) return done(rc);
        if ((rc = sl.feed.reserve((size_t)off))) return done(rc);
        FEED_TRY(hipMemcpy(sl.chrs.p, chrs.data(), sizeof(ChrDev) * (size_t)p->nchr, hipMemcpyHostToDevice));
        if (!items.empty()) FEED_TRY(hipMemcpy(sl.items.p, items.data(), sizeof(FeedItem) * items.size(), hipMemcpyHostToDevice));
        FEED_TRY(hipMemset(sl.counter.p, 0, 2 * sizeof(int32_t)));
    }
    p->plan.valid = false;
    // ---- every size's chain kernel, each on its own stream; every element of a feed is written by its kernel
    for (int i = 0; i < n_sizes; i++) {
        garlic_panel::FeedSlot &sl = *p->feed_slots[(size_t)i];
        FEED_TRY(hipEventRecord(sl.ev0, sl.stream));
        if (n_items[(size_t)i]) {
            FeedArgs f{p->d_packed.p, p->d_tab.p, sl.items.p, sl.chrs.p, sl.feed.p, ind_idx ? d_rowmap.p : nullptr, p->nwordrows, 0,
                       p->nind, winsizes[i], (int32_t)n_items[(size_t)i], steps[i], getenv("GARLIC_FEED_NO_ASM") ? 0 : 1,
                       sl.counter.p, nullptr};
            void *kargs[] = {(void *)&f};
            FEED_TRY(hipLaunchKernel((const void *)lod_feed_kernel, dim3((unsigned)grids[(size_t)i]), dim3(FEED_G * WAVE), kargs, 0, sl.stream));
        }
        FEED_TRY(hipEventRecord(sl.ev1, sl.stream));
        FEED_TRY(hipGetLastError());
    }
    // ---- the feeds, in order
    for (int i = 0; i < n_sizes; i++) {
        garlic_panel::FeedSlot &sl = *p->feed_slots[(size_t)i];
        if (n_items[(size_t)i])
            FEED_TRY(hipMemcpyAsync(feeds[i], sl.feed.p, sizeof(double) * (size_t)total[(size_t)i], hipMemcpyDeviceToHost, sl.stream));
    }
    float ms_sum = 0.f;
    for (int i = 0; i < n_sizes; i++) {
        garlic_panel::FeedSlot &sl = *p->feed_slots[(size_t)i];
        FEED_TRY(hipStreamSynchronize(sl.stream));
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, sl.ev0, sl.ev1) == hipSuccess) ms_sum += ms;
    }
    p->stats = garlic_call_stats{};
    p->stats.chain_kernel_ms = ms_sum;    // (the sizes overlap: the sum of their spans, not wall time)
    p->stats_pending = false;
    return done(GARLIC_OK);
#undef FEED_TRY
}

// Coverage counts of the unweighted --error scores without the scores: chain + compare + sliding count in one kernel
// (coverage_kernel.hpp).  Where that kernel does not apply -- a cutoff at or below MISSING, terms that are not all
// finite, a window sum that can be -9999.0, W > COVF_MAX_W -- the scores are computed into the panel's scratch and
// counted by garlic_roh_coverage.
static int coverage_bits_layout(garlic_panel *p, std::vector<ChrDev> &bchrs, std::vector<int32_t> &word_base, int64_t &total)
{
    bchrs.assign((size_t)p->nchr, ChrDev{});
    word_base.assign((size_t)p->nchr + 1, 0);
    total = 0;
    for (int c = 0; c < p->nchr; c++) {
        const int64_t words = (p->chr_nloci[c] + 31) / 32;
        bchrs[(size_t)c] = ChrDev{p->chr_off[c], total, words, p->chr_nloci[c], 0};
        total += words * p->nind;
        word_base[(size_t)c + 1] = word_base[(size_t)c] + (int32_t)words;
        if (words * 4 * (int64_t)p->nind >= (int64_t)1 << 32) return fail(GARLIC_ERR_INVALID, "chromosome %d: bit rows beyond 32-bit offsets", c);
    }
    return GARLIC_OK;
}

// What becomes of the window bits: the sliding counts (garlic_roh_coverage_fused) or the ROH segments (garlic_roh_segments)
struct CovSink {
    int16_t *inwin = nullptr;            // counts
    int32_t inwin_pitch_align = 8, where = GARLIC_DEVICE;
    bool segments = false;               // segments
    double overlap_frac = 0.0;
    garlic_roh_segment *segs = nullptr;
    int64_t cap = 0, *n_out = nullptr;
};

// ROH segments from the window bits (roh_segments_kernel.hpp): r bits, break bits, the list; sorted on the host into
// the reference's order.  The bit matrix is [chromosome][individual][word], bchrs / word_base as coverage_bits_layout.
static int segments_from_bits(garlic_panel *p, const uint32_t *d_bits, const ChrDev *d_bchrs, const std::vector<ChrDev> &bchrs,
                              const std::vector<int32_t> &word_base, int32_t W, const CovSink &sink)
{
    garlic_ctx *ctx = p->ctx;
    hipStream_t s = ctx->stream;
    double T = sink.overlap_frac * W;              // src/garlic-roh.cpp:421-423
    T = (T >= 1) ? T : 1;
    T = (T <= W) ? T : W;
    const int thr = (int)std::ceil(T);             // counts are integers: cnt >= T  <=>  cnt >= ceil(T)
    const size_t nchr = (size_t)p->nchr;
    const int64_t total_words = word_base[nchr];
    int64_t bit_words = 0;
    for (size_t c = 0; c < nchr; c++) bit_words = std::max<int64_t>(bit_words, bchrs[c].out_base + bchrs[c].out_pitch * p->nind);
    PoolBuf<uint32_t> d_mask;
    PoolBuf<garlic_roh_segment> d_segs;
    DevBuf<uint32_t> d_brk;
    DevBuf<int32_t> d_wbase;
    DevBuf<unsigned long long> d_count;
    DevBuf<int32_t> d_w0, d_wedge_chr;             // chromosomes whose first SNP is at position 0 (roh_segments_kernel.hpp)
    std::vector<int32_t> wedge_chr;
    for (int c = 0; c < p->nchr; c++)
        if (p->chr_nloci[c] > 0 && p->pos[(size_t)p->chr_off[c]] == 0) wedge_chr.push_back(c);
    auto done = [&](int code) { d_mask.release(); d_brk.release(); d_wbase.release(); d_segs.release(); d_count.release();
                                d_w0.release(); d_wedge_chr.release(); return code; };
    int rc;
    const int64_t cap = std::max<int64_t>(sink.cap, 0);
    if ((rc = d_mask.reserve(ctx, (size_t)std::max<int64_t>(bit_words, 1))) || (rc = d_brk.reserve((size_t)std::max<int64_t>(total_words, 1))) ||
        (rc = d_wbase.reserve(word_base.size())) || (rc = d_segs.reserve(ctx, (size_t)std::max<int64_t>(cap, 1))) || (rc = d_count.reserve(1)))
        return done(rc);
    hipError_t e = hipMemcpyAsync(d_wbase.p, word_base.data(), sizeof(int32_t) * word_base.size(), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_brk.p, 0, sizeof(uint32_t) * (size_t)std::max<int64_t>(total_words, 1), s);
    if (e == hipSuccess) e = hipMemsetAsync(d_count.p, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "roh segments: %s", hipGetErrorString(e)));
    unsigned long long found = 0;
    if (total_words > 0) {
        const int nb = (int)p->boundaries.size();
        if (nb > 0)
            hipLaunchKernelGGL(roh_break_bits_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, p->d_boundaries.p, nb,
                               p->d_chr_off.p, d_wbase.p, p->nchr, d_brk.p);
        const dim3 grid((unsigned)((total_words + 255) / 256), (unsigned)((p->nind + ROH_ROWS - 1) / ROH_ROWS));
        hipLaunchKernelGGL(roh_mask_from_bits_kernel, grid, dim3(256), 0, s, d_bits, d_bchrs, d_wbase.p, p->nchr, p->nind, W, thr, d_mask.p);
        const int32_t *a_w0 = nullptr;
        if (!wedge_chr.empty()) {
            const size_t n_w0 = nchr * (size_t)p->nind;
            if ((rc = d_w0.reserve(n_w0)) || (rc = d_wedge_chr.reserve(wedge_chr.size()))) return done(rc);
            e = hipMemsetAsync(d_w0.p, 0xff, sizeof(int32_t) * n_w0, s);            // -1: an ordinary row
            if (e == hipSuccess)
                e = hipMemcpyAsync(d_wedge_chr.p, wedge_chr.data(), sizeof(int32_t) * wedge_chr.size(), hipMemcpyHostToDevice, s);
            if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "roh segments: %s", hipGetErrorString(e)));
            const int n_threads = (int)wedge_chr.size() * p->nind;
            hipLaunchKernelGGL(roh_wedge_kernel, dim3((unsigned)((n_threads + 63) / 64)), dim3(64), 0, s, d_mask.p, d_bchrs, d_brk.p, d_wbase.p,
                               d_wedge_chr.p, (int)wedge_chr.size(), p->nind, T, d_w0.p, d_segs.p, (long long)cap, d_count.p);
            a_w0 = d_w0.p;
        }
        hipLaunchKernelGGL(roh_segments_from_mask_kernel, grid, dim3(256), 0, s, d_mask.p, d_bchrs, d_brk.p, d_wbase.p, p->nchr, p->nind, T,
                           d_segs.p, (long long)cap, d_count.p, a_w0);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(&found, d_count.p, sizeof found, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "roh segments: %s", hipGetErrorString(e)));
    }
    if (sink.n_out) *sink.n_out = (int64_t)found;
    if ((int64_t)found <= cap && found > 0) {
        e = hipMemcpyAsync(sink.segs, d_segs.p, sizeof(garlic_roh_segment) * found, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "roh segments: %s", hipGetErrorString(e)));
        std::sort(sink.segs, sink.segs + found, [](const garlic_roh_segment &x, const garlic_roh_segment &y) {
            return x.ind != y.ind ? x.ind < y.ind : x.chr != y.chr ? x.chr < y.chr : x.start < y.start;
        });
    }
    return done(GARLIC_OK);
}

static int coverage_impl(garlic_panel *p, int32_t winsize, double error, int32_t max_gap, int32_t use_gl,
                         int32_t weighted, int32_t M, double mu, double cutoff, const CovSink &sink)
{
    int16_t *const inwin = sink.inwin;
    const int32_t inwin_pitch_align = sink.inwin_pitch_align, where = sink.where;
    garlic_ctx *ctx = p->ctx;
    int rc;
    if ((rc = set_device(ctx))) return rc;
    if (!p->have_map || !p->have_freq || !p->have_geno)
        return fail(GARLIC_ERR_STATE, "panel needs map, freq and genotypes before computing LOD");
    if ((rc = ensure_segments(p, max_gap))) return rc;
    if ((rc = ensure_term_table(p, error))) return rc;
    const int32_t W = winsize;
    auto unfused = [&]() -> int {      // scores into the panel's scratch, then garlic_roh_coverage
        const Layout L = make_layout(p, 32, p->nind);
        int rc2;
        if ((rc2 = p->d_out.reserve(ctx, (size_t)L.total))) return rc2;
        if (weighted) rc2 = garlic_wlod_windows(p, W, error, max_gap, use_gl, M, mu, 0, p->nind, 32, p->d_out.p, GARLIC_DEVICE);
        else rc2 = garlic_lod_windows(p, W, error, max_gap, use_gl, 0, p->nind, 32, p->d_out.p, GARLIC_DEVICE);
        if (rc2) return rc2;
        if (!sink.segments) return garlic_roh_coverage(p, p->d_out.p, 32, p->nind, W, cutoff, inwin, inwin_pitch_align, where);
        // segments: the scores' bits (score >= cutoff, MISSING compared like any score), then as from the chains' bits
        std::vector<ChrDev> bchrs, schrs((size_t)p->nchr);
        std::vector<int32_t> word_base;
        int64_t boff = 0;
        if ((rc2 = coverage_bits_layout(p, bchrs, word_base, boff))) return rc2;
        for (int c = 0; c < p->nchr; c++) schrs[(size_t)c] = ChrDev{p->chr_off[c], L.base[c], L.pitch[c], p->chr_nloci[c], 0};
        PoolBuf<uint32_t> d_bits;
        DevBuf<ChrDev> d_bchrs, d_schrs;
        DevBuf<int32_t> d_wbase;
        auto done = [&](int code) { d_bits.release(); d_bchrs.release(); d_schrs.release(); d_wbase.release(); return code; };
        if ((rc2 = d_bits.reserve(ctx, (size_t)std::max<int64_t>(boff, 1))) || (rc2 = d_bchrs.reserve(bchrs.size())) ||
            (rc2 = d_schrs.reserve(schrs.size())) || (rc2 = d_wbase.reserve(word_base.size())))
            return done(rc2);
        hipStream_t s2 = ctx->stream;
        hipError_t e2 = hipMemcpyAsync(d_bchrs.p, bchrs.data(), sizeof(ChrDev) * bchrs.size(), hipMemcpyHostToDevice, s2);
        if (e2 == hipSuccess) e2 = hipMemcpyAsync(d_schrs.p, schrs.data(), sizeof(ChrDev) * schrs.size(), hipMemcpyHostToDevice, s2);
        if (e2 == hipSuccess) e2 = hipMemcpyAsync(d_wbase.p, word_base.data(), sizeof(int32_t) * word_base.size(), hipMemcpyHostToDevice, s2);
        if (e2 != hipSuccess) return done(fail(GARLIC_ERR_HIP, "roh segments: %s", hipGetErrorString(e2)));
        if (word_base[(size_t)p->nchr] > 0)
            hipLaunchKernelGGL(roh_bits_from_scores_kernel, dim3((unsigned)((word_base[(size_t)p->nchr] + 255) / 256), (unsigned)p->nind),
                               dim3(256), 0, s2, p->d_out.p, d_schrs.p, d_bchrs.p, d_wbase.p, p->nchr, W, cutoff, d_bits.p);
        e2 = hipStreamSynchronize(s2);      // (the host vectors above)
        if (e2 != hipSuccess) return done(fail(GARLIC_ERR_HIP, "roh segments: %s", hipGetErrorString(e2)));
        return done(segments_from_bits(p, d_bits.p, d_bchrs.p, bchrs, word_base, W, sink));
    };
    hipStream_t s = ctx->stream;
    if ((weighted || use_gl) && cutoff > MISSING_D && !getenv("GARLIC_COVERAGE_UNFUSED")) {
        // --weighted (with or without likelihoods): the tuned wLOD kernels leave 16 bits per individual and group instead
        // of 16 scores (wlod_write_group), the counts come from the bits as for the unweighted scores
        std::vector<ChrDev> bchrs;
        std::vector<int32_t> word_base;
        int64_t boff = 0;
        if ((rc = coverage_bits_layout(p, bchrs, word_base, boff))) return rc;
        const Layout Lo = make_layout(p, inwin_pitch_align, p->nind);
        std::vector<ChrDev> ochrs((size_t)p->nchr);
        for (int c = 0; c < p->nchr; c++) ochrs[(size_t)c] = ChrDev{p->chr_off[c], Lo.base[c], Lo.pitch[c], p->chr_nloci[c], 0};
        PoolBuf<uint32_t> d_bits;
        DevBuf<ChrDev> d_bchrs, d_ochrs;
        DevBuf<int32_t> d_wbase;
        DevBuf<int16_t> d_cov;
        auto done = [&](int code) { d_bits.release(); d_bchrs.release(); d_ochrs.release(); d_wbase.release(); d_cov.release(); return code; };
        if ((rc = d_bits.reserve(ctx, (size_t)std::max<int64_t>(boff, 4))) || (rc = d_bchrs.reserve(bchrs.size())) ||
            (rc = d_ochrs.reserve(ochrs.size())) || (rc = d_wbase.reserve(word_base.size())))
            return done(rc);
        int16_t *dst = inwin;
        if (where == GARLIC_HOST) {
            if ((rc = d_cov.reserve((size_t)Lo.total))) return done(rc);
            dst = d_cov.p;
        }
        hipError_t e = hipMemcpyAsync(d_bchrs.p, bchrs.data(), sizeof(ChrDev) * bchrs.size(), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(d_ochrs.p, ochrs.data(), sizeof(ChrDev) * ochrs.size(), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(d_wbase.p, word_base.data(), sizeof(int32_t) * word_base.size(), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemsetAsync(d_bits.p, 0, sizeof(uint32_t) * (size_t)std::max<int64_t>(boff, 4), s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);      // (the host vectors above)
        if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "coverage: %s", hipGetErrorString(e)));
        p->cov_pending = CovBits{d_bits.p, d_bchrs.p, cutoff};
        p->cov_written = false;
        if (weighted)
            rc = garlic_wlod_windows(p, W, error, max_gap, use_gl, M, mu, 0, p->nind, 32, reinterpret_cast<double *>(d_bits.p), GARLIC_DEVICE);
        else      // unweighted scores with likelihoods: the TGLS ring chain leaves a dword of bits per lane and tile
            rc = garlic_lod_windows(p, W, error, max_gap, 1, 0, p->nind, 32, reinterpret_cast<double *>(d_bits.p), GARLIC_DEVICE);
        const bool written = weighted || p->cov_written;
        p->cov_pending = CovBits{nullptr, nullptr, 0.0};
        if (rc == GARLIC_INTERNAL_NO_BITS) return done(unfused());
        if (rc) return done(rc);
        if (!written) return done(unfused());       // (no scored window at all: nothing was launched)
        if (sink.segments) return done(segments_from_bits(p, d_bits.p, d_bchrs.p, bchrs, word_base, W, sink));
        bool vec_ok = (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
        for (int c = 0; c < p->nchr; c++) vec_ok = vec_ok && Lo.base[c] % 8 == 0 && Lo.pitch[c] % 8 == 0;
        hipLaunchKernelGGL(cov_counts_from_bits_kernel, dim3((unsigned)((word_base[(size_t)p->nchr] + 255) / 256),
                                                             (unsigned)((p->nind + COV_ITEM_ROWS - 1) / COV_ITEM_ROWS)),
                           dim3(256), 0, s, d_bits.p, d_bchrs.p, d_ochrs.p, d_wbase.p, p->nchr, W, p->nind, vec_ok ? 1 : 0, dst);
        e = hipGetLastError();
        if (e == hipSuccess && where == GARLIC_HOST)
            e = hipMemcpyAsync(inwin, dst, sizeof(int16_t) * (size_t)Lo.total, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "coverage: %s", hipGetErrorString(e)));
        return done(GARLIC_OK);
    }
    const bool fused = !weighted && !use_gl && W <= COVF_MAX_W && cutoff > MISSING_D && p->tab_all_finite &&
                       !lod_exact_needed(p, MODE_LOD, W) && !getenv("GARLIC_COVERAGE_UNFUSED");
    if (!fused) return unfused();
    const int nblk = (p->nind + WAVE - 1) / WAVE;
    std::vector<Run> runs;
    std::vector<FillItem> fill;
    int64_t n_valid = 0;
    plan_runs(p, W, runs, fill, n_valid);              // in chromosome and position order
    std::vector<int> order(runs.size());
    for (size_t k = 0; k < runs.size(); k++) order[k] = (int)k;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return (runs[x].b - runs[x].a) > (runs[y].b - runs[y].a); });
    std::vector<int32_t> col0(runs.size(), 0);
    std::vector<FeedItem> items;
    build_feed_items(runs, order, nullptr, nblk, col0, items);
    const Layout Lo = make_layout(p, inwin_pitch_align, p->nind);
    std::vector<ChrDev> chrs((size_t)p->nchr);
    for (int c = 0; c < p->nchr; c++) chrs[(size_t)c] = ChrDev{p->chr_off[c], Lo.base[c], Lo.pitch[c], p->chr_nloci[c], 0};
    DevBuf<FeedItem> d_items;
    DevBuf<ChrDev> d_chrs;
    DevBuf<int32_t> d_counter;
    DevBuf<int16_t> d_cov;
    auto done = [&](int code) { d_items.release(); d_chrs.release(); d_counter.release(); d_cov.release(); return code; };
    if ((rc = d_items.reserve(std::max<size_t>(items.size(), 1))) || (rc = d_chrs.reserve(chrs.size())) ||
        (rc = d_counter.reserve(4)))
        return done(rc);
    int16_t *dst = inwin;
    if (where == GARLIC_HOST) {
        if ((rc = d_cov.reserve((size_t)Lo.total))) return done(rc);
        dst = d_cov.p;
    }
    hipError_t e = hipMemcpyAsync(d_chrs.p, chrs.data(), sizeof(ChrDev) * chrs.size(), hipMemcpyHostToDevice, s);
    if (e == hipSuccess && !items.empty())
        e = hipMemcpyAsync(d_items.p, items.data(), sizeof(FeedItem) * items.size(), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_counter.p, 0, 4 * sizeof(int32_t), s);
    if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "coverage: %s", hipGetErrorString(e)));
    bool vec_ok = (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
    for (int c = 0; c < p->nchr; c++) vec_ok = vec_ok && Lo.base[c] % 8 == 0 && Lo.pitch[c] % 8 == 0;
    const int slot = (int)(ctx->n_calls % garlic_ctx::HIST);
    {
        // two kernels: one bit per window and individual from the hand-scheduled chain (lod_bits_kernel), then the
        // sliding counts from the bits (cov_counts_from_bits_kernel): a 64th of the score bytes in between
        std::vector<ChrDev> bchrs((size_t)p->nchr);
        std::vector<int32_t> word_base((size_t)p->nchr + 1, 0);
        int64_t boff = 0;
        for (int c = 0; c < p->nchr; c++) {
            const int64_t words = (p->chr_nloci[c] + 31) / 32;
            // rows a multiple of eight dwords long: the chain kernel stores eight tiles' dwords as one aligned 32-byte piece
            const int64_t row_words = (words + 7) / 8 * 8;
            bchrs[(size_t)c] = ChrDev{p->chr_off[c], boff, row_words, p->chr_nloci[c], 0};
            boff += row_words * p->nind;
            word_base[(size_t)c + 1] = word_base[(size_t)c] + (int32_t)words;
            if (row_words * 4 * (int64_t)p->nind >= (int64_t)1 << 32)
                return done(fail(GARLIC_ERR_INVALID, "chromosome %d: bit rows beyond 32-bit offsets", c));
        }
        PoolBuf<uint32_t> d_bits;
        DevBuf<ChrDev> d_bchrs;
        DevBuf<int32_t> d_wbase;
        auto done2 = [&](int code) { d_bits.release(); d_bchrs.release(); d_wbase.release(); return done(code); };
        if ((rc = d_bits.reserve(ctx, (size_t)std::max<int64_t>(boff, 1))) || (rc = d_bchrs.reserve(bchrs.size())) ||
            (rc = d_wbase.reserve(word_base.size())))
            return done2(rc);
        e = hipMemcpyAsync(d_bchrs.p, bchrs.data(), sizeof(ChrDev) * bchrs.size(), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(d_wbase.p, word_base.data(), sizeof(int32_t) * word_base.size(), hipMemcpyHostToDevice, s);
        (void)hipEventRecord(ctx->hist0[slot], s);
        if (e == hipSuccess) e = hipMemsetAsync(d_bits.p, 0, sizeof(uint32_t) * (size_t)boff, s);
        if (e != hipSuccess) return done2(fail(GARLIC_ERR_HIP, "coverage: %s", hipGetErrorString(e)));
        // GARLIC_COVERAGE_OVERLAP=1: the counts ride in the chain kernel's queue (FeedArgs::cnt_order) instead of a launch of
        // their own behind it -- the longest runs' chains are that kernel's critical path and leave most of the chip idle,
        // the counts of every finished chromosome could fill it.  Built and measured ("Coverage
        // counts without the scores"): the chains are latency-bound and the counts' traffic slows the longest one from
        // 22.8 to 36.5 ns per window, 19.5 ms against 17.6 ms for the two launches at 10M x 1250.  Off by default.
        const bool overlap = !items.empty() && !sink.segments && getenv("GARLIC_COVERAGE_OVERLAP");
        DevBuf<int32_t> d_cnt;      // chr_done[nchr] | timeout | chr_need[nchr] | cnt_order[nchr] | cnt_base[nchr + 1]
        auto done3 = [&](int code) { d_cnt.release(); return done2(code); };
        const size_t nchr = (size_t)p->nchr;
        int64_t n_cnt_items = 0;
        if (overlap) {
            std::vector<int32_t> h(4 * nchr + 2, 0);
            int32_t *need = h.data() + nchr + 1, *corder = need + nchr, *cbase = corder + nchr;
            std::vector<int32_t> longest(nchr, 0);
            for (const FeedItem &it : items) need[it.chr]++;
            for (const Run &r : runs) longest[(size_t)r.chr] = std::max(longest[(size_t)r.chr], r.b - r.a + 1);
            for (size_t c = 0; c < nchr; c++) corder[c] = (int32_t)c;
            std::stable_sort(corder, corder + nchr, [&](int32_t x, int32_t y) { return longest[(size_t)x] < longest[(size_t)y]; });
            const int64_t nrg = (p->nind + COV_ITEM_ROWS - 1) / COV_ITEM_ROWS;
            for (size_t k = 0; k < nchr; k++) {
                cbase[k] = (int32_t)n_cnt_items;
                const int64_t words = (p->chr_nloci[corder[k]] + 31) / 32;
                n_cnt_items += (words + COV_ITEM_WORDS - 1) / COV_ITEM_WORDS * nrg;
            }
            cbase[nchr] = (int32_t)n_cnt_items;
            if (n_cnt_items + (int64_t)items.size() >= ((int64_t)1 << 31))
                return done3(fail(GARLIC_ERR_INVALID, "coverage: more than 2^31 work items"));
            if ((rc = d_cnt.reserve(h.size()))) return done3(rc);
            e = hipMemcpyAsync(d_cnt.p, h.data(), sizeof(int32_t) * h.size(), hipMemcpyHostToDevice, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);      // (h leaves scope)
            if (e != hipSuccess) return done3(fail(GARLIC_ERR_HIP, "coverage: %s", hipGetErrorString(e)));
        }
        if (!items.empty()) {
            FeedArgs f{p->d_packed.p, p->d_tab.p, d_items.p, d_bchrs.p, reinterpret_cast<double *>(d_bits.p), nullptr, p->nwordrows, 0,
                       p->nind, W, (int32_t)items.size(), 1, getenv("GARLIC_FEED_NO_ASM") ? 0 : 1, d_counter.p, nullptr, cutoff};
            if (overlap) {
                f.chr_done = d_cnt.p;
                f.cnt_timeout = d_cnt.p + nchr;
                f.chr_need = d_cnt.p + nchr + 1;
                f.cnt_order = d_cnt.p + 2 * nchr + 1;
                f.cnt_base = d_cnt.p + 3 * nchr + 1;
                f.cnt_chrs = d_chrs.p;
                f.cnt_out = dst;
                f.n_cnt_items = (int32_t)n_cnt_items;
                f.n_cnt_chr = p->nchr;
                f.cnt_vec_ok = vec_ok ? 1 : 0;
            }
            DevBuf<int64_t> d_ftrace;   // debugging aid: GARLIC_TRACE=<file> dumps the chain items' time stamps
            const char *ftrace_path = getenv("GARLIC_TRACE");
            if (ftrace_path && d_ftrace.reserve(8 * items.size()) == GARLIC_OK) {
                (void)hipMemsetAsync(d_ftrace.p, 0, sizeof(int64_t) * 8 * items.size(), s);
                f.trace = d_ftrace.p;
            }
            // workgroups per CU as the chains want them (feed_grid: the same chains, the same paces); the count items behind
            // them in the queue are short and fill whatever is idle.  (Persistent workgroups, all resident: a count item
            // that waits must not keep a chain item from starting.)
            int per_cu = 1, chain_grid = 1;
            if ((rc = feed_grid(ctx, items, &chain_grid, &per_cu))) return done3(rc);
            const int grid = (int)std::min<size_t>(items.size() + (size_t)n_cnt_items, (size_t)ctx->n_cu * per_cu);
            void *kargs[] = {(void *)&f};
            e = hipLaunchKernel((const void *)lod_bits_kernel, dim3((unsigned)grid), dim3(FEED_G * WAVE), kargs, 0, s);
            if (e != hipSuccess) return done3(fail(GARLIC_ERR_HIP, "coverage: %s", hipGetErrorString(e)));
            if (f.trace) {
                std::vector<int64_t> tr(8 * items.size());
                (void)hipMemcpyAsync(tr.data(), d_ftrace.p, sizeof(int64_t) * tr.size(), hipMemcpyDeviceToHost, s);
                (void)hipStreamSynchronize(s);
                if (FILE *fo = fopen(ftrace_path, "w")) {
                    for (size_t i = 0; i < items.size(); i++) {
                        fprintf(fo, "%zu", i);
                        for (int q = 0; q < 8; q++) fprintf(fo, " %lld", (long long)tr[8 * i + q]);
                        fprintf(fo, "\n");
                    }
                    fclose(fo);
                }
                d_ftrace.release();
            }
        }
        if (sink.segments) {
            (void)hipEventRecord(ctx->hist1[slot], s);
            ctx->n_calls++;
            return done3(segments_from_bits(p, d_bits.p, d_bchrs.p, bchrs, word_base, W, sink));
        }
        if (!overlap)
            hipLaunchKernelGGL(cov_counts_from_bits_kernel, dim3((unsigned)((word_base[(size_t)p->nchr] + 255) / 256),
                                                                 (unsigned)((p->nind + COV_ITEM_ROWS - 1) / COV_ITEM_ROWS)),
                               dim3(256), 0, s, d_bits.p, d_bchrs.p, d_chrs.p, d_wbase.p, p->nchr, W, p->nind, vec_ok ? 1 : 0, dst);
        (void)hipEventRecord(ctx->hist1[slot], s);
        ctx->n_calls++;
        e = hipGetLastError();
        if (e == hipSuccess && where == GARLIC_HOST)
            e = hipMemcpyAsync(inwin, dst, sizeof(int16_t) * (size_t)Lo.total, hipMemcpyDeviceToHost, s);
        int32_t timed_out = 0;
        if (e == hipSuccess && overlap)
            e = hipMemcpyAsync(&timed_out, d_cnt.p + nchr, sizeof(int32_t), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return done3(fail(GARLIC_ERR_HIP, "coverage: %s", hipGetErrorString(e)));
        if (timed_out) {
            p->n_count_timeouts++;
            return done3(fail(GARLIC_ERR_HIP, "coverage: a count item gave up waiting for its chromosome's chains"));
        }
        return done3(GARLIC_OK);
    }
}

int garlic_roh_coverage_fused(garlic_panel *p, int32_t winsize, double error, int32_t max_gap, int32_t use_gl,
                              int32_t weighted, int32_t M, double mu, double cutoff, int16_t *inwin,
                              int32_t inwin_pitch_align, int32_t where)
{
    if (!p || !inwin) return fail(GARLIC_ERR_INVALID, "panel and inwin are required");
    if (winsize <= 1 || inwin_pitch_align < 1) return fail(GARLIC_ERR_INVALID, "winsize must be > 1, inwin_pitch_align >= 1");
    if (winsize > 32767) return fail(GARLIC_ERR_INVALID, "coverage counts are 16-bit: winsize <= 32767");
    if (p->nind > 65535) return fail(GARLIC_ERR_INVALID, "coverage: at most 65535 individuals per call");
    CovSink sink;
    sink.inwin = inwin;
    sink.inwin_pitch_align = inwin_pitch_align;
    sink.where = where;
    return coverage_impl(p, winsize, error, max_gap, use_gl, weighted, M, mu, cutoff, sink);
}

int garlic_roh_segments(garlic_panel *p, int32_t winsize, double error, int32_t max_gap, int32_t use_gl, int32_t weighted,
                        int32_t M, double mu, double cutoff, double overlap_frac, garlic_roh_segment *segments,
                        int64_t capacity, int64_t *n_segments)
{
    if (!p || !n_segments) return fail(GARLIC_ERR_INVALID, "panel and n_segments are required");
    if (capacity < 0 || (capacity > 0 && !segments)) return fail(GARLIC_ERR_INVALID, "segments: capacity without a buffer");
    if (winsize <= 1) return fail(GARLIC_ERR_INVALID, "winsize must be > 1");
    if (winsize > 32767) return fail(GARLIC_ERR_INVALID, "coverage counts are 16-bit: winsize <= 32767");
    if (p->nind > 65535) return fail(GARLIC_ERR_INVALID, "coverage: at most 65535 individuals per call");
    if (!(overlap_frac == overlap_frac)) return fail(GARLIC_ERR_INVALID, "overlap_frac is not a number");
    *n_segments = 0;
    // (the reference tells "a segment is open" by its first position being > 0 and "none" by < 0, src/garlic-roh.cpp:456, 493,
    // 514: a chromosome that starts at 0 takes the device's restatement of what that does; a negative position has no meaning)
    if (p->have_map)
        for (int c = 0; c < p->nchr; c++)
            if (p->chr_nloci[c] > 0 && p->pos[(size_t)p->chr_off[c]] < 0)
                return fail(GARLIC_ERR_INVALID, "chromosome %d starts at position %d: ROH segments need positions >= 0", c,
                            (int)p->pos[(size_t)p->chr_off[c]]);
    CovSink sink;
    sink.segments = true;
    sink.overlap_frac = overlap_frac;
    sink.segs = segments;
    sink.cap = capacity;
    sink.n_out = n_segments;
    return coverage_impl(p, winsize, error, max_gap, use_gl, weighted, M, mu, cutoff, sink);
}

int garlic_panel_tgls_mode(garlic_panel *p, int32_t *mode, int32_t *terms_by)
{
    if (!p || !mode) return fail(GARLIC_ERR_INVALID, "panel and mode are required");
    *mode = !p->have_gl ? 0 : (p->gl_cont ? GARLIC_TGLS_CONTINUOUS : GARLIC_TGLS_DICTIONARY);
    if (terms_by) *terms_by = p->glterms_valid ? p->gl_terms_by : 0;
    return GARLIC_OK;
}

int garlic_roh_coverage(garlic_panel *p, const double *scores, int32_t pitch_align, int32_t nind_out,
                        int32_t winsize, double cutoff, int16_t *inwin, int32_t inwin_pitch_align,
                        int32_t where)
{
    if (!p || !scores || !inwin) return fail(GARLIC_ERR_INVALID, "panel, scores and inwin are required");
    if (winsize <= 1 || pitch_align < 1 || inwin_pitch_align < 1 || nind_out < 1)
        return fail(GARLIC_ERR_INVALID, "winsize must be > 1; pitch_align, inwin_pitch_align and nind_out >= 1");
    if (winsize > 32767) return fail(GARLIC_ERR_INVALID, "coverage counts are 16-bit: winsize <= 32767");
    if (nind_out > 65535) return fail(GARLIC_ERR_INVALID, "coverage: at most 65535 individuals per call");
    int rc;
    if ((rc = set_device(p->ctx))) return rc;
    hipStream_t s = p->ctx->stream;
    const Layout L = make_layout(p, pitch_align, nind_out), Lo = make_layout(p, inwin_pitch_align, nind_out);
    std::vector<ChrDev> chrs(2 * (size_t)p->nchr);
    std::vector<int32_t> seg_base(p->nchr + 1, 0);
    for (int c = 0; c < p->nchr; c++) {
        chrs[c] = ChrDev{p->chr_off[c], L.base[c], L.pitch[c], p->chr_nloci[c], 0};
        chrs[p->nchr + c] = ChrDev{p->chr_off[c], Lo.base[c], Lo.pitch[c], p->chr_nloci[c], 0};
        seg_base[c + 1] = seg_base[c] + (p->chr_nloci[c] + COV_SEG - 1) / COV_SEG;
    }
    DevBuf<ChrDev> d_chrs;
    DevBuf<int32_t> d_seg;
    DevBuf<int16_t> d_cov;
    auto done = [&](int code) { d_chrs.release(); d_seg.release(); d_cov.release(); return code; };
    if ((rc = d_chrs.reserve(chrs.size())) || (rc = d_seg.reserve(seg_base.size()))) return done(rc);
    int16_t *dst = inwin;
    if (where == GARLIC_HOST) {
        if ((rc = d_cov.reserve((size_t)Lo.total))) return done(rc);
        dst = d_cov.p;
    }
    hipError_t e = hipMemcpyAsync(d_chrs.p, chrs.data(), sizeof(ChrDev) * chrs.size(), hipMemcpyHostToDevice, s);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_seg.p, seg_base.data(), sizeof(int32_t) * seg_base.size(), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "coverage: %s", hipGetErrorString(e)));
    const size_t lds = sizeof(uint16_t) * ((size_t)winsize + COV_SEG + 2);      // (counts <= COV_SEG + W - 1: 16 bits, W < 57000)
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(roh_coverage_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "coverage: %s", hipGetErrorString(e)));
    bool vec_ok = (reinterpret_cast<uintptr_t>(dst) & 15) == 0;        // eight counts per store: every row 16-B aligned
    for (int c = 0; c < p->nchr && vec_ok; c++) vec_ok = Lo.base[c] % 8 == 0 && Lo.pitch[c] % 8 == 0;
    hipLaunchKernelGGL(roh_coverage_kernel, dim3((unsigned)seg_base[p->nchr], (unsigned)nind_out),
                       dim3(COV_THREADS), lds, s, scores, d_chrs.p, d_chrs.p + p->nchr, d_seg.p, p->nchr, nind_out,
                       winsize, cutoff, dst, vec_ok ? 1 : 0);
    e = hipGetLastError();
    if (e == hipSuccess && where == GARLIC_HOST)
        e = hipMemcpyAsync(inwin, dst, sizeof(int16_t) * (size_t)Lo.total, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return done(fail(GARLIC_ERR_HIP, "coverage: %s", hipGetErrorString(e)));
    return done(GARLIC_OK);
}

int garlic_last_call_stats(garlic_panel *p, garlic_call_stats *stats)
{
    if (!p || !stats) return fail(GARLIC_ERR_INVALID, "panel and stats are required");
    if (p->stats_pending) {   // the event times of the last call (waits for it if it is still running)
        int rc;
        if ((rc = set_device(p->ctx))) return rc;
        HIP_TRY(hipEventSynchronize(p->ctx->ev_end));
        (void)hipEventElapsedTime(&p->stats.chain_kernel_ms, p->ctx->hist0[p->stats_slot], p->ctx->hist1[p->stats_slot]);
        (void)hipEventElapsedTime(&p->stats.total_ms, p->ctx->ev_begin, p->ctx->ev_end);
        p->stats_pending = false;
    }
    if (p->d_counter.p) {      // strip launches the tile form had to repair (counted on the device; waits for the stream)
        int32_t n = 0;
        int rc;
        if ((rc = set_device(p->ctx))) return rc;
        HIP_TRY(hipMemcpyAsync(&n, p->d_counter.p + 4, sizeof(int32_t), hipMemcpyDeviceToHost, p->ctx->stream));
        HIP_TRY(hipStreamSynchronize(p->ctx->stream));
        p->stats.n_stall_reruns = n;
    }
    p->stats.n_count_timeouts = p->n_count_timeouts;
    *stats = p->stats;
    return GARLIC_OK;
}

int garlic_recent_kernel_ms(garlic_ctx *ctx, float *ms, int32_t n, int32_t *got)
{
    if (!ctx || !ms || !got || n < 1) return fail(GARLIC_ERR_INVALID, "ctx, ms, got are required; n >= 1");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const int64_t have = std::min<int64_t>(std::min<int64_t>(n, garlic_ctx::HIST), ctx->n_calls);
    for (int64_t k = 0; k < have; k++) {       // oldest of the requested ones first
        const int64_t call = ctx->n_calls - have + k;
        ms[k] = 0.f;
        (void)hipEventElapsedTime(&ms[k], ctx->hist0[call % garlic_ctx::HIST], ctx->hist1[call % garlic_ctx::HIST]);
    }
    *got = (int32_t)have;
    return GARLIC_OK;
}

int garlic_device_alloc(garlic_ctx *ctx, int64_t bytes, void **out)
{
    if (!ctx || !out || bytes < 1) return fail(GARLIC_ERR_INVALID, "context, size and result pointer are required");
    int rc;
    if ((rc = set_device(ctx))) return rc;
    return score_alloc(ctx, (size_t)bytes, out);
}

int garlic_device_free(garlic_ctx *ctx, void *ptr)
{
    if (!ctx) return fail(GARLIC_ERR_INVALID, "context is required");
    if (!ptr) return GARLIC_OK;
    int rc;
    if ((rc = set_device(ctx))) return rc;
    return score_free(ctx, ptr);
}

int garlic_device_trim(garlic_ctx *ctx)
{
    if (!ctx) return fail(GARLIC_ERR_INVALID, "context is required");
    int rc;
    if ((rc = set_device(ctx))) return rc;
    return score_pool_trim();
}

int garlic_device_alloc_stats(garlic_ctx *ctx, int64_t *live_bytes, int64_t *pooled_bytes, int64_t *reserved_bytes)
{
    if (!ctx) return fail(GARLIC_ERR_INVALID, "context is required");
    std::lock_guard<std::mutex> lock(g_score_mutex);
    int64_t live = 0, pooled = 0;
    for (const ScoreAlloc &a : g_score_allocs)
        if (a.device == ctx->device) (a.pooled ? pooled : live) += (int64_t)a.size;
    if (live_bytes) *live_bytes = live;
    if (pooled_bytes) *pooled_bytes = pooled;
    if (reserved_bytes) *reserved_bytes = live + pooled + g_score_retired[ctx->device < 16 ? ctx->device : 15];
    return GARLIC_OK;
}

// Score memory in the chain kernel's fast placement: `candidates` buffers side by side, the real
// kernel for `winsize` timed into each (one pass that builds the plan, four in a row, the last three timed), the
// fastest kept, the others returned to the pool.
int garlic_panel_alloc_scores(garlic_panel *p, int32_t pitch_align, int32_t nind_out, int32_t winsize, double error,
                              int32_t max_gap, int32_t candidates, void **out, float *candidate_ms)
{
    if (!p || !out) return fail(GARLIC_ERR_INVALID, "panel and result pointer are required");
    if (nind_out < 1 || nind_out > p->nind) return fail(GARLIC_ERR_INVALID, "nind_out outside the panel");
    int rc;
    if ((rc = set_device(p->ctx))) return rc;
    *out = nullptr;
    if (candidates <= 0) candidates = 4;
    candidates = std::min(candidates, 16);
    const Layout L = make_layout(p, pitch_align, nind_out);
    // Candidates come in rounds.  Buffers of one round are cut from neighbouring physical memory and can ALL land on
    // the slow side (profiles/r03_bench_plain_all_candidates_slow.json: eight candidates at 1.65 ms on a fresh device, 1.35 ms
    // one process later), and the times are not two clean classes either (BENCH_r03: 1.44 kept / 1.50 median / 1.54 worst
    // in one round, 1.34-1.40 on other leases), so a relative spread inside a round says little.  The criterion is the
    // kernel's own bound: a round whose best candidate takes its score bytes at >= 0.74 of the HBM peak (1.39 ms at 1M SNPs
    // x 1000 individuals) has found the fast placement; otherwise another round is taken from fresh memory while the
    // earlier ones are still held (so that the allocator cannot hand the same pages back) -- three rounds at most
    // (GARLIC_ALLOC_ROUNDS), 2 s at most, memory permitting.  The best of all rounds is kept; how many were drawn and what
    // they timed: garlic_panel_alloc_scores_info.
    int max_rounds = 3;
    if (const char *e = getenv("GARLIC_ALLOC_ROUNDS")) max_rounds = std::max(1, std::min(atoi(e), 4));
    const size_t bytes = sizeof(double) * (size_t)L.total;
    const float target_ms = (float)((double)bytes / (0.74 * 8.0e12) * 1e3);
    std::vector<void *> cand;
    std::vector<float> ms;
    std::vector<size_t> round_start;
    auto cleanup = [&](int keep) {
        for (int k = 0; k < (int)cand.size(); k++)
            if (k != keep && cand[(size_t)k]) (void)score_free(p->ctx, cand[(size_t)k]);
    };
    garlic_ctx *ctx = p->ctx;
    const bool was_async = ctx->async_device;
    const auto t_begin = std::chrono::steady_clock::now();
    int best = -1, best_round = 0;
    for (int round = 0; round < max_rounds; round++) {
        const size_t base = cand.size();
        round_start.push_back(base);
        for (int k = 0; k < candidates; k++) {
            void *q = nullptr;
            if ((rc = score_alloc(ctx, bytes, &q))) break;
            cand.push_back(q);
            ms.push_back(0.f);
        }
        const bool short_round = rc != 0;
        if (short_round) {
            if (round == 0 && cand.empty()) return rc;
            g_last_error.clear();   // no room for (all of) another round: what there is stands
            rc = 0;
        }
        // passes enqueued back to back, as a caller that keeps the scores on the device issues them (a pass that is waited
        // for runs ~5 % faster than one in a queue: DESIGN.md section 4): one pass to build the plan, then four in a row,
        // the last three timed by their HIP events
        for (size_t k = base; k < cand.size(); k++) {
            ctx->async_device = false;
            rc = launch_lod(p, MODE_LOD, winsize, error, max_gap, 0, 0.0, 0, nind_out, pitch_align, (double *)cand[k], GARLIC_DEVICE);
            ctx->async_device = true;
            for (int pass = 0; pass < 4 && !rc; pass++)
                rc = launch_lod(p, MODE_LOD, winsize, error, max_gap, 0, 0.0, 0, nind_out, pitch_align, (double *)cand[k], GARLIC_DEVICE);
            ctx->async_device = was_async;
            if (rc) { cleanup(-1); return rc; }
            hipError_t e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) { cleanup(-1); return fail(GARLIC_ERR_HIP, "alloc_scores: %s", hipGetErrorString(e)); }
            float acc = 0.f;
            for (int q = 1; q <= 3; q++) {
                const int slot = (int)((ctx->n_calls - q) % garlic_ctx::HIST);
                float t = 0.f;
                (void)hipEventElapsedTime(&t, ctx->hist0[slot], ctx->hist1[slot]);
                acc += t;
            }
            ms[k] = acc / 3;
            if (best < 0 || ms[k] < ms[(size_t)best]) { best = (int)k; best_round = round; }
        }
        const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        if (short_round || ms[(size_t)best] <= target_ms || elapsed > 2.0) break;
    }
    if (candidate_ms) {
        memset(candidate_ms, 0, sizeof(float) * (size_t)candidates);
        const size_t r0 = round_start[(size_t)best_round];
        const size_t r1 = (size_t)best_round + 1 < round_start.size() ? round_start[(size_t)best_round + 1] : ms.size();
        memcpy(candidate_ms, ms.data() + r0, sizeof(float) * std::min((size_t)candidates, r1 - r0));
    }
    {
        std::vector<float> sorted(ms);
        std::sort(sorted.begin(), sorted.end());
        p->placement = garlic_panel::Placement{(int32_t)ms.size(), (int32_t)round_start.size(), sorted.front(),
                                               sorted[sorted.size() / 2], sorted.back(), target_ms,
                                               sorted.front() <= target_ms ? 1 : 0};
    }
    cleanup(best);
    *out = cand[(size_t)best];
    return GARLIC_OK;
}

int garlic_panel_alloc_scores_info(garlic_panel *p, int32_t *drawn, int32_t *rounds, float *best_ms, float *median_ms,
                                   float *worst_ms, float *target_ms, int32_t *reached_target)
{
    if (!p) return fail(GARLIC_ERR_INVALID, "panel is NULL");
    if (p->placement.drawn == 0) return fail(GARLIC_ERR_STATE, "garlic_panel_alloc_scores has not run on this panel");
    if (drawn) *drawn = p->placement.drawn;
    if (rounds) *rounds = p->placement.rounds;
    if (best_ms) *best_ms = p->placement.best_ms;
    if (median_ms) *median_ms = p->placement.median_ms;
    if (worst_ms) *worst_ms = p->placement.worst_ms;
    if (target_ms) *target_ms = p->placement.target_ms;
    if (reached_target) *reached_target = p->placement.reached;
    return GARLIC_OK;
}

int garlic_panel_chain_kind(garlic_panel *p, int32_t *kind)
{
    if (!p || !kind) return fail(GARLIC_ERR_INVALID, "panel and kind are required");
    *kind = p->last_chain_kind;
    return GARLIC_OK;
}

int garlic_ctx_set_async(garlic_ctx *ctx, int32_t on)
{
    if (!ctx) return fail(GARLIC_ERR_INVALID, "ctx is NULL");
    ctx->async_device = on != 0;
    return GARLIC_OK;
}

} // extern "C"
