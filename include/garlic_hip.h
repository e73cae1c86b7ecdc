/*
 * garlic_hip.h -- C ABI of libgarlic_hip.so: GARLIC Phase-I window LOD scores on MI355X (gfx950).
 *
 * This is the drop-in boundary for the one hot path this project replaces.  The reference
 * (szpiech/garlic v1.1.6a) has no FFI layer; the boundary is the pair of C++ entry points
 *
 *     calcLODWindows   src/garlic-roh.h:96-102   (body src/garlic-roh.cpp:279-309, calcLOD :18-132)
 *     calcwLODWindows  src/garlic-roh.h:104-112  (body src/garlic-roh.cpp:311-347, calcwLOD :144-277)
 *
 * and the structs they borrow (src/garlic-data.h:32-108).  Every function below takes plain
 * pointers and sizes; the C++ adapter in garlic_amd/host mirrors the reference signatures on top
 * (INTEGRATION.md shows the few lines a GARLIC maintainer would add).
 *
 * Conventions
 *   - Every call returns GARLIC_OK (0) or a GARLIC_ERR_* code; garlic_hip_last_error() gives the
 *     text for the calling thread.  The reference throws `int 0` (src/garlic-data.cpp:1619); the
 *     adapter converts non-zero to `throw 0`.
 *   - `where` says whether a caller buffer is host (GARLIC_HOST) or device (GARLIC_DEVICE) memory.
 *   - Loci of all chromosomes are concatenated in file order ("global locus index"); chromosome c
 *     owns [chr_off[c], chr_off[c+1]).
 *   - Genotypes are the reference's codes (src/garlic-data.cpp:109-129): 0,1,2 = copies of the
 *     counted allele, anything else (-9) = missing.
 *   - LOD output is individual-major like WinData::data (src/garlic-data.h:83): element
 *     (chr c, ind i, locus l) lives at  chr_base[c] + i*chr_pitch[c] + l  (in doubles), see
 *     garlic_lod_out_layout().  Windows that hold no score are exactly -9999.0
 *     (MISSING, src/garlic-data.h:24); every element is written by the call, no pre-fill needed.
 *   - Results are bit-identical to the reference's doubles on the same inputs and host libm.
 *   - A context is bound to one device and one HIP stream; calls on one context must not overlap
 *     in time, different contexts are independent (one per GPU for individual sharding).
 */
#ifndef GARLIC_HIP_H
#define GARLIC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: the LD functions take `phased`; garlic_panel_set_phase
 * 3: likelihoods may be continuous (no 256-value limit), garlic_panel_tgls_mode; an LD subsample may be
 *    empty (sub_idx != NULL, n_sub = 0); garlic_lod_feed_subset
 * 4: garlic_device_alloc / garlic_device_free (score matrices), garlic_panel_chain_kind
 * 5: garlic_lod_feed_multi (the feeds of several window sizes in one call); garlic_panel_alloc_scores,
 *    garlic_device_alloc_stats, garlic_device_trim
 * 6: garlic_roh_coverage_fused (coverage counts without the score matrix)
 * 7: garlic_roh_segments (the ROH segments of assembleROHWindows without scores or counts) */
#define GARLIC_HIP_ABI_VERSION 8

#define GARLIC_OK 0
#define GARLIC_ERR_INVALID 1  /* bad argument (e.g. winsize <= 1: src/garlic-cli.cpp:433-442) */
#define GARLIC_ERR_HIP 2      /* HIP runtime failure / no gfx950 device */
#define GARLIC_ERR_STATE 3    /* inputs missing for the requested computation */
#define GARLIC_ERR_NOMEM 4

#define GARLIC_HOST 0
#define GARLIC_DEVICE 1

#define GARLIC_MISSING (-9999.0)

#define GARLIC_GL_GQ 0 /* --gl-type GQ, src/garlic-data.cpp:1557 */
#define GARLIC_GL_GL 1 /* --gl-type GL, src/garlic-data.cpp:1562 */
#define GARLIC_GL_PL 2 /* --gl-type PL, src/garlic-data.cpp:1566 */

typedef struct garlic_ctx garlic_ctx;
typedef struct garlic_panel garlic_panel;

int garlic_hip_abi_version(void);
const char *garlic_hip_last_error(void);
int garlic_hip_device_count(int32_t *count);

/* Context: device ordinal + stream.  hip_stream == NULL: the context creates its own stream;
 * otherwise a hipStream_t owned by the caller (e.g. torch's current stream). */
int garlic_ctx_create(int32_t device, void *hip_stream, garlic_ctx **ctx);
int garlic_ctx_destroy(garlic_ctx *ctx);
int garlic_ctx_synchronize(garlic_ctx *ctx);
/* By default every call returns after its device work has finished.  With on != 0,
 * garlic_lod_windows / garlic_wlod_windows calls whose output stays on the device and whose
 * arguments repeat the previous call's (same window size, range, layout: the work plan is reused)
 * only enqueue their kernels on the context's stream; garlic_ctx_synchronize (or
 * garlic_last_call_stats, or any synchronous call) waits.  For callers that issue passes back to
 * back (benchmarks, pipelines that consume the scores on the same stream). */
int garlic_ctx_set_async(garlic_ctx *ctx, int32_t on);

/* Device memory for score matrices (the `out` of garlic_lod_windows & co. with GARLIC_DEVICE): any device pointer
 * works as `out`; garlic_device_alloc / garlic_device_free / garlic_panel_alloc_scores (below, by the other score
 * calls) hand out memory the unweighted kernel runs fastest into.  GARLIC itself has no counterpart (WinData rows
 * are host memory, garlic-data.cpp:1690); keep the buffer across window sizes as GARLIC keeps its WinData. */
/* HIP-event durations (ms) of the dominant kernel of the context's most recent window-score calls,
 * oldest first, at most 32: lets a caller time asynchronous passes without waiting for each.  Waits
 * for the stream.  *got = number of values written (<= n). */
int garlic_recent_kernel_ms(garlic_ctx *ctx, float *ms, int32_t n, int32_t *got);

/* Panel = what calcLODWindows borrows: HapData / MapData / FreqData (/ GenoLikeData / LDData)
 * of every chromosome, for the nind individuals this context owns (a shard of the TFAM order).
 * chr_nloci[c] = MapData::nloci of chromosome c (src/garlic-data.h:58). */
int garlic_panel_create(garlic_ctx *ctx, int32_t nchr, const int32_t *chr_nloci, int32_t nind,
                        garlic_panel **panel);
int garlic_panel_destroy(garlic_panel *panel);

/* MapData::physicalPos / geneticPos (src/garlic-data.h:53-54) concatenated over chromosomes
 * (host arrays), and centromere::centromereStart/End per chromosome
 * (src/garlic-roh.cpp:36-37; unknown chromosome => 0,0).  gpos may be NULL (unweighted). */
int garlic_panel_set_map(garlic_panel *panel, const int32_t *pos, const double *gpos,
                         const int32_t *centro_start, const int32_t *centro_end);

/* FreqData::freq concatenated (host array, src/garlic-data.h:71). */
int garlic_panel_set_freq(garlic_panel *panel, const double *freq);

/* HapData::data rows (src/garlic-data.h:35) for global loci [locus_begin, locus_begin+locus_count):
 * geno[(l - locus_begin) * ld + i] is individual i of this shard (ld >= nind lets a shard read a
 * column block of the full matrix).  May be called repeatedly to stream a panel in chunks.
 * Device side: 2-bit packed, 16 loci per 32-bit word, individual-minor. */
int garlic_panel_set_genotypes(garlic_panel *panel, const int16_t *geno, int64_t ld,
                               int64_t locus_begin, int64_t locus_count, int32_t where);

/* The same genotypes from SNP-major 2-bit rows -- what a genotype cache or a bed-like reader holds:
 * 4 genotypes per byte, codes 0/1/2 = HapData::data values, 3 = missing (-9).  Row l - locus_begin
 * starts at rows + (l - locus_begin) * row_bytes and holds the individuals of the WHOLE data set;
 * this shard's individual i is number ind_offset + i of the row (byte / 4, bits 2 * (% 4)).
 * An eighth of the host memory and PCIe traffic of the int16 rows; same device layout. */
int garlic_panel_set_genotypes_2bit(garlic_panel *panel, const uint8_t *rows, int64_t row_bytes,
                                    int64_t ind_offset, int64_t locus_begin, int64_t locus_count,
                                    int32_t where);

/* GenoLikeData::data (src/garlic-data.h:91): per-genotype error probabilities, already converted
 * as readTGLSData does (src/garlic-data.cpp:1557-1576); same addressing as genotypes.  Any doubles:
 * while a panel has seen at most 256 distinct values (--gl-type GQ, PL integers) it keeps one-byte
 * dictionary codes and the host tabulates lod() per (SNP, value, genotype) with the host libm; beyond
 * that (--gl-type GL, continuous values) it keeps the values themselves (8 bytes per genotype) and
 * lod() (src/garlic-roh.cpp:355-386) runs on the device with glibc's log10 restated operation by
 * operation -- checked against the host's log10 when first needed; should they ever differ, the terms
 * are computed on the host instead.  Same scores either way: those of the reference on this host.
 * When the device cannot hold values and terms side by side the values are converted in place at the
 * first computation; changing genotypes, frequencies, the map or the weighting parameters (M, mu)
 * afterwards then needs the likelihoods uploaded again, over all loci (GARLIC_ERR_STATE says so). */
int garlic_panel_set_gl(garlic_panel *panel, const double *gl, int64_t ld, int64_t locus_begin,
                        int64_t locus_count, int32_t where);

/* The same likelihoods already dictionary-coded, as a reader that converts GQ / PL integers produces
 * them anyway: codes[(l - locus_begin) * ld + i] indexes values[0 .. nvalues), nvalues <= 256, the
 * error probabilities as readTGLSData converts them.  One byte per genotype on the host and over
 * PCIe instead of eight.  Every call may bring its own table; a panel whose tables add up to more than
 * 256 distinct values keeps the values themselves from then on (see garlic_panel_set_gl). */
int garlic_panel_set_gl_codes(garlic_panel *panel, const uint8_t *codes, int64_t ld, int64_t locus_begin,
                              int64_t locus_count, const double *values, int32_t nvalues, int32_t where);

/* How the panel holds its likelihoods: 0 none yet, GARLIC_TGLS_DICTIONARY, GARLIC_TGLS_CONTINUOUS.
 * *terms_by (may be NULL): who computed the current TGLS term matrix -- 0 nothing computed yet or
 * tabulated from the dictionary, 1 the device (log10 verified against the host), 2 the host. */
#define GARLIC_TGLS_DICTIONARY 1
#define GARLIC_TGLS_CONTINUOUS 2
int garlic_panel_tgls_mode(garlic_panel *panel, int32_t *mode, int32_t *terms_by);

/* HapData::firstCopy (src/garlic-data.h:36; filled by readTPED under --phased,
 * src/garlic-data.cpp:106,129: "the first allele of the pair is the counted allele"), one byte per
 * genotype, non-zero = true; same addressing as genotypes.  Only the phased LD weights read it. */
int garlic_panel_set_phase(garlic_panel *panel, const uint8_t *first_copy, int64_t ld,
                           int64_t locus_begin, int64_t locus_count, int32_t where);

/* LDData::LD (src/garlic-data.h:105) for winsize: ld[l * winsize + k], l global locus. */
int garlic_panel_set_ld(garlic_panel *panel, int32_t winsize, const double *ld, int32_t where);

/* calcLDData (src/garlic-data.cpp:330-375): the LD weights of wLOD, computed on the device from the
 * panel's genotypes,
 *     LD[s][k] = sum_{i = s .. s+winsize-1, in order} (i == s+k ? 1 : c(i, s+k))   for s <= nloci_c - winsize
 * with  phased == 0:  c = hr2 (calcHR2LD :377-527, hr2 :558-583), from homFreq (:656-676) taken over
 *                     every individual of the panel;
 *       phased != 0:  c = r2  (calcR2LD :426-535, r2 :585-617; --phased), from the allele
 *                     frequencies of garlic_panel_set_freq and the phase of garlic_panel_set_phase.
 * The pair counts run over the individuals sub_idx[0 .. n_sub); sub_idx == NULL: all of them; a
 * non-NULL sub_idx with n_sub = 0: none (what a shard that holds no member of a panel-wide subsample
 * passes to garlic_ld_counts: its pair counts are zero, its locus counts still cover everyone).  The
 * reference draws that subsample with a time-seeded RNG (:346-364, --ld-subsample); here the caller
 * supplies it.  The result is installed as the panel's LD weights for winsize (as garlic_panel_set_ld would)
 * and, if ld_out is not NULL, also written there (nloci * winsize doubles, ld[l * winsize + k]). */
int garlic_panel_compute_ld(garlic_panel *panel, int32_t winsize, int32_t phased, const int32_t *sub_idx,
                            int32_t n_sub, double *ld_out, int32_t where);

/* The same in two steps for panels whose individuals are sharded over GPUs: everything that
 * depends on genotypes is an integer count, exact and order-free --
 *     locus_counts[l][2]            = {homozygous, non-missing} individuals of this shard
 *     pair_counts[l][winsize][2]    = over the shard's part of the subsample, for SNP pairs (l, l+d), d >= 1:
 *                                     {both non-missing, both non-missing and homozygous}   (hr2), or
 *                                     {2 * both non-missing, x11 of r2 :592-606}            (phased)
 * -- so the caller sums the count arrays of all shards element-wise (one all-reduce) and every
 * shard finishes with the summed counts; the floating-point part then runs in the reference's
 * operation order and is identical on every GPU.  sub_idx is shard-local.  (Phased: the allele
 * frequencies are the caller's, over all individuals, as for the LOD scores.) */
int garlic_ld_counts(garlic_panel *panel, int32_t winsize, int32_t phased, const int32_t *sub_idx,
                     int32_t n_sub, int32_t *locus_counts, int32_t *pair_counts, int32_t where);
int garlic_ld_finish(garlic_panel *panel, int32_t winsize, int32_t phased, const int32_t *locus_counts,
                     const int32_t *pair_counts, double *ld_out, int32_t where);

/* A panel keeps its device scratch between calls (LD counting and summing buffers: about
 * 5 x nloci x winsize x 8 bytes; the score scratch of host-output and feed calls), because at scale
 * allocating it costs more than the kernels.  This frees it; inputs, tables, the installed LD weights
 * and the TGLS term matrix stay, and the next call allocates again what it needs. */
int garlic_panel_release_scratch(garlic_panel *panel);

/* Output addressing for this panel: pitch_align = 1 gives the reference's dense rows
 * (chr_pitch[c] = chr_nloci[c], chromosome blocks back to back).  A larger value rounds every row
 * pitch and chromosome base up to that many doubles AND reserves rows up to the next multiple of
 * 64 individuals per chromosome block (32 = 256-byte aligned rows: the layout the tuned kernel
 * needs; with pitch_align = 1 a slower generic kernel path runs).  Pad rows / pad columns are
 * never read back and hold unspecified values.  total = number of doubles the caller's buffer
 * must have for nind_out individuals. */
int garlic_lod_out_layout(garlic_panel *panel, int32_t pitch_align, int32_t nind_out,
                          int64_t *chr_base, int64_t *chr_pitch, int64_t *total);

/* calcLODWindows (src/garlic-roh.cpp:279): unweighted window LOD scores of individuals
 * [ind_begin, ind_begin + ind_count) for one window size.  use_gl != 0 takes the per-genotype
 * error from the panel's GL data (USE_GL, src/garlic-roh.cpp:68) instead of `error`.
 * out has garlic_lod_out_layout(pitch_align, ind_count) doubles. */
int garlic_lod_windows(garlic_panel *panel, int32_t winsize, double error, int32_t max_gap,
                       int32_t use_gl, int32_t ind_begin, int32_t ind_count, int32_t pitch_align,
                       double *out, int32_t where);

/* The same for several window sizes (--winsize-multi; exploreWinsizes, src/garlic-roh.cpp:726-751):
 * the scores of winsizes[k] go to out + k * out_stride (out_stride >= the layout's total).  The
 * panel stays resident, the kernels run one window size after the other. */
int garlic_lod_windows_multi(garlic_panel *panel, const int32_t *winsizes, int32_t n_winsizes, double error,
                             int32_t max_gap, int32_t use_gl, int32_t ind_begin, int32_t ind_count,
                             int32_t pitch_align, double *out, int64_t out_stride, int32_t where);

/* calcwLODWindows (src/garlic-roh.cpp:311): gap-weighted wLOD; needs gpos and LD for winsize
 * (garlic_panel_set_ld or garlic_panel_compute_ld).  winsize up to 4096 (above that:
 * GARLIC_ERR_INVALID; the reference has no limit but no use for such windows either). */
int garlic_wlod_windows(garlic_panel *panel, int32_t winsize, double error, int32_t max_gap,
                        int32_t use_gl, int32_t M, double mu, int32_t ind_begin, int32_t ind_count,
                        int32_t pitch_align, double *out, int32_t where);

/* convertWinData2DoubleData (src/garlic-data.cpp:2026-2069) on the device: the KDE feed.  Reads
 * scores laid out as garlic_lod_out_layout(pitch_align, nind_out) describes (device memory) and
 * writes, in the reference's order chromosome -> individual -> locus, every `step`-th window
 * (locus 0, step, 2*step, ...) that is neither MISSING nor NaN.  feed (device) must hold
 * feed_capacity doubles; *count (host) receives the number written (or needed, if larger than the
 * capacity -- nothing is written then).  The explore / auto-winsize flows keep only these values
 * (step = winsize), 8/winsize bytes per window instead of 8. */
int garlic_lod_flatten(garlic_panel *panel, const double *scores, int32_t pitch_align, int32_t nind_out,
                       int32_t step, double *feed, int64_t feed_capacity, int64_t *count);

/* The explore / auto-winsize flows (src/garlic-roh.cpp:726-751, 798-837, 881-920) compute the
 * scores of a window size only to thin them into the KDE feed and throw them away.  This is both
 * steps in one call with the scores kept on the device: garlic_lod_windows or garlic_wlod_windows
 * (weighted != 0; needs LD for winsize) over every individual of the panel, then
 * convertWinData2DoubleData with `step` (garlic_lod_flatten).  feed: HOST buffer of feed_capacity
 * doubles; *count = values produced (nothing is copied if it exceeds the capacity: an upper bound is
 * sum_c ceil(nloci_c / step) * nind); chr_counts (may be NULL): values per chromosome, so that
 * feeds of individual shards can be merged in the reference's chromosome -> individual order.
 * 8 / step bytes per window cross PCIe instead of 8.  Unweighted --error scores with step >= 4 are
 * thinned by the LOD kernel itself (only the sampled windows are ever stored: 8 / step bytes per
 * window of HBM writes, no full-size scratch); the other variants compute the full scores into
 * device scratch and sample them there.  Same values either way. */
int garlic_lod_feed(garlic_panel *panel, int32_t winsize, double error, int32_t max_gap, int32_t use_gl,
                    int32_t weighted, int32_t M, double mu, int32_t step, double *feed,
                    int64_t feed_capacity, int64_t *count, int64_t *chr_counts);

/* convertSubsetWinData2DoubleData (src/garlic-data.cpp:2071-2150; selectLODCutoff, src/garlic-roh.cpp:
 * 674-675, --kde-subsample): garlic_lod_feed for the individuals ind_idx[0 .. n_idx) only, in that
 * order inside every chromosome (chromosome -> ind_idx[0] .. ind_idx[n_idx-1] -> locus).  The reference
 * draws them with a time-seeded gsl_ran_choose, which keeps the TFAM order; here the caller supplies
 * them (distinct, each in [0, nind)).  Only the 64-individual blocks that hold a listed individual
 * are scored.  ind_idx == NULL: everyone (= garlic_lod_feed).  chr_counts as there. */
int garlic_lod_feed_subset(garlic_panel *panel, int32_t winsize, double error, int32_t max_gap, int32_t use_gl,
                           int32_t weighted, int32_t M, double mu, int32_t step, const int32_t *ind_idx,
                           int32_t n_idx, double *feed, int64_t feed_capacity, int64_t *count, int64_t *chr_counts);

/* The callers that sweep window sizes -- exploreWinsizes (src/garlic-roh.cpp:726-751), selectWinsize (:798-837),
 * selectWinsizeFromList (:881-920: --winsize-multi with --auto-winsize) -- run calcLODWindows + the KDE thinning once
 * per size on the same data.  This is garlic_lod_feed_subset for n_sizes window sizes in one call (unweighted --error
 * scores; steps[i] is the thinning step of winsizes[i], the reference uses the window size).  Every size gets a
 * stream and device scratch of its own and all of them are enqueued before the first feed is fetched: the tail of one
 * size's chain kernel (its longest runs, a few waves per CU) runs beside the bulk of the next size's, and a feed crosses
 * PCIe while the following sizes are computed.  feeds[i]: HOST buffer of feed_capacity[i] doubles; counts[i] as
 * *count there; chr_counts (may be NULL): [n_sizes][nchr].  ind_idx / n_idx as in garlic_lod_feed_subset (NULL:
 * everyone).  Values identical to n_sizes single calls. */
int garlic_lod_feed_multi(garlic_panel *panel, const int32_t *winsizes, const int32_t *steps, int32_t n_sizes, double error,
                          int32_t max_gap, const int32_t *ind_idx, int32_t n_idx, double *const *feeds,
                          const int64_t *feed_capacity, int64_t *counts, int64_t *chr_counts);

/* First half of assembleROHWindows (src/garlic-roh.cpp:446-454) on the device: for every individual
 * and SNP the number of windows with score >= cutoff that cover the SNP,
 *     inWin[l] = #{ w in (l - winsize, l] : scores[w] >= cutoff }
 * (MISSING and NaN never qualify; the reference indexes past the array for a qualifying window in
 * the last winsize-1 positions, which only a cutoff <= -9999 can produce: here such windows cover
 * the SNPs that exist).  scores: device memory laid out as garlic_lod_out_layout(pitch_align,
 * nind_out); inwin: int16 elements addressed the same way with inwin_pitch_align (1 = dense rows of
 * nloci_c), host or device per `where`.  The caller compares inWin with
 * OVERLAP_FRAC * winsize clamped to [1, winsize] (:422-424) -- 2 bytes per (individual, SNP) cross
 * PCIe instead of 8. */
int garlic_roh_coverage(garlic_panel *panel, const double *scores, int32_t pitch_align, int32_t nind_out,
                        int32_t winsize, double cutoff, int16_t *inwin, int32_t inwin_pitch_align,
                        int32_t where);

/* The same counts for the unweighted --error scores of every individual of the panel, computed WITHOUT the scores:
 * calcLOD's chain (src/garlic-roh.cpp:18-132) leaves one bit per window and individual -- score >= cutoff, the score
 * itself only ever in a register -- and the inWin[] loop (:446-454) becomes a count over the last winsize bits.  What
 * GARLIC's final pass needs when --raw-lod is not asked for: 2 bytes per window leave the device and no score matrix
 * is resident.  inwin as for garlic_roh_coverage (inwin_pitch_align a multiple of 8 lets the kernel store 16 bytes
 * at a time).  use_gl / weighted / M / mu as for garlic_lod_windows / garlic_wlod_windows: with `weighted` the tuned
 * wLOD kernels leave the bits themselves (16 per individual and group instead of 16 scores), with unweighted
 * per-genotype likelihoods the TGLS chain does (32 per individual and tile).  Falls back to scores + garlic_roh_coverage where the bit form
 * does not apply (cutoff <= -9999; unweighted: winsize > 1024, non-finite terms, window sums that can be -9999.0). */
int garlic_roh_coverage_fused(garlic_panel *panel, int32_t winsize, double error, int32_t max_gap, int32_t use_gl,
                              int32_t weighted, int32_t M, double mu, double cutoff, int16_t *inwin,
                              int32_t inwin_pitch_align, int32_t where);

/* The ROH segments themselves: assembleROHWindows (src/garlic-roh.cpp:409-545) from the panel to its
 * rohData->start / stop lists, with neither the scores nor the coverage counts ever in memory.  The window bits come as
 * for garlic_roh_coverage_fused (same arguments, same fallbacks); on the device they become "SNP is covered by at least
 * OVERLAP_FRAC * winsize qualifying windows" bits (threshold clamped to [1, winsize], :421-423), and the four-branch
 * walk over every individual's SNPs (:456-533) becomes: maximal stretches of such SNPs, cut where two neighbours are
 * more than max_gap apart or straddle the centromere, kept when they hold at least the threshold's number of SNPs
 * (and not begun at the chromosome's last SNP: the reference never closes such a segment).
 *   segments[k] = {individual (panel-relative), chromosome (index in the panel), first SNP, last SNP}, SNP indices
 *   chromosome-local and inclusive; the caller maps them to positions (physicalPos / geneticPos of :470-520).
 *   Ordered by individual, chromosome, first SNP -- the order the reference appends them in.
 * *n_segments is the number found; when it exceeds `capacity` nothing usable is in `segments` (call again with room;
 * capacity 0 / segments NULL just counts).  A few MB for a 10M-SNP x 1250-individual shard, against 25 GB of counts.
 * Positions must be >= 0.  The reference tells "a segment is open" by its first POSITION being > 0 and "none" by
 * < 0 (:456, :493, :514): on a 0-based map a stretch opened at a chromosome's SNP 0 is neither -- nothing closes it but
 * a covered SNP behind a break, and it is reported from SNP 0 to the SNP in front of that one whatever lies between.
 * Reproduced as is (pinned against the real assembleROHWindows). */
typedef struct garlic_roh_segment {
    int32_t ind, chr, start, stop;
} garlic_roh_segment;
int garlic_roh_segments(garlic_panel *panel, int32_t winsize, double error, int32_t max_gap, int32_t use_gl, int32_t weighted,
                        int32_t M, double mu, double cutoff, double overlap_frac, garlic_roh_segment *segments,
                        int64_t capacity, int64_t *n_segments);

/* Introspection used by tests and the bench (device work of the last garlic_*_windows call). */
typedef struct garlic_call_stats {
    int64_t n_segments;      /* gap/centromere-free SNP segments over all chromosomes */
    int64_t n_runs;          /* segments long enough to hold a window (maximal valid runs) */
    int64_t n_chain_items;   /* (run, 64-individual block) work items = wavefronts launched */
    int64_t n_valid_windows; /* scored windows per individual */
    int64_t n_missing;       /* MISSING windows per individual */
    float chain_kernel_ms;   /* HIP-event time of the dominant kernel on the context stream */
    float total_ms;          /* HIP-event time of the whole call's device work */
    /* ABI 8: liveness book-keeping, cumulative since the panel was created; both are expected to stay 0.
     * n_stall_reruns: launches of the strip kernel (--weighted with per-genotype likelihoods) in which a wave ran out of
     * its poll budget and that the tile form, enqueued behind every strip launch, therefore recomputed (on the device:
     * no call synchronises for it).  n_count_timeouts: garlic_roh_coverage_fused calls that failed because a count item
     * gave up waiting for its chromosome's chains (GARLIC_COVERAGE_OVERLAP=1 only). */
    int64_t n_stall_reruns;
    int64_t n_count_timeouts;
} garlic_call_stats;
int garlic_last_call_stats(garlic_panel *panel, garlic_call_stats *stats);

/* Which rolling-sum chain the last unweighted / TGLS score call ran.  The reference decides "the previous window
 * holds no score" by VALUE (garlic-roh.cpp:79), so a scored window that sums to exactly -9999.0 restarts the sum.
 * 0: such a sum is impossible for this panel and window size (W x the most negative term stays above -9999): the
 * tuned chain; 1: possible -- the tuned chain ran, its scored windows were scanned, none was -9999.0; 2: one was
 * (or the environment forces it): the chain that follows the reference to the letter ran (11-17 x slower). */
int garlic_panel_chain_kind(garlic_panel *panel, int32_t *kind);

/* Score memory for where = GARLIC_DEVICE calls.  garlic_device_alloc: a virtual range backed by physical chunks of
 * its own (HIP virtual memory management; hipMalloc where the driver has none).  Freed buffers stay mapped in a pool
 * and are handed out again for requests they fit (a caller that allocates per sweep reuses the same few ranges; the
 * pool is capped at a quarter of the device memory, GARLIC_ALLOC_POOL_GB).  garlic_device_alloc_stats: bytes handed
 * out, bytes idle in the pool, bytes of address space reserved in all (any of the three may be NULL).
 * Where a score buffer sits in VRAM decides between two speeds of the unweighted kernel (1.36 / 1.62 ms at 1M SNPs x
 * 1000 individuals).  garlic_panel_alloc_scores allocates `candidates` (0 = 4) buffers for the
 * layout garlic_lod_out_layout(pitch_align, nind_out), times the real kernel for `winsize` into each and keeps the
 * fastest; candidate_ms (may be NULL): the kernel time into each candidate.  Buffers allocated together can all sit on
 * the slow side, so unless the best candidate of a round takes its score bytes at >= 0.74 of the HBM peak (the fast
 * placement; out of reach for small panels, which then simply use the budget) a further round is taken from fresh memory
 * while the earlier ones are held: three rounds at most (GARLIC_ALLOC_ROUNDS), 2 s at most, memory permitting;
 * candidate_ms then holds the times of the round the kept buffer came from.  garlic_panel_alloc_scores_info (ABI 8):
 * how many candidates the last call on this panel drew in how many rounds, the best / median / worst kernel time over
 * all of them, the time the 0.74 target corresponds to and whether the kept buffer reached it (any pointer may be NULL).
 * Free with garlic_device_free.  The library's own full-score scratch (host-output calls) is chosen the same way at first
 * use. */
int garlic_device_alloc(garlic_ctx *ctx, int64_t bytes, void **out);
int garlic_device_free(garlic_ctx *ctx, void *ptr);
int garlic_device_alloc_stats(garlic_ctx *ctx, int64_t *live_bytes, int64_t *pooled_bytes, int64_t *reserved_bytes);
int garlic_device_trim(garlic_ctx *ctx);   /* idle pooled buffers give their memory back now (the library does this itself
                                              when one of its own allocations runs out of memory) */
int garlic_panel_alloc_scores(garlic_panel *panel, int32_t pitch_align, int32_t nind_out, int32_t winsize, double error,
                              int32_t max_gap, int32_t candidates, void **out, float *candidate_ms);
int garlic_panel_alloc_scores_info(garlic_panel *panel, int32_t *drawn, int32_t *rounds, float *best_ms, float *median_ms,
                                   float *worst_ms, float *target_ms, int32_t *reached_target);

#ifdef __cplusplus
}
#endif
#endif /* GARLIC_HIP_H */
