/* TEST INFRASTRUCTURE ONLY -- see lod_oracle.h.
 *
 * Plain-C restatement of GARLIC v1.1.6a Phase I.  Every function names the reference
 * lines it follows (paths under /root/reference/src).  Built with the reference's own
 * flags (-O3 -m64 -msse2, no FMA) plus -ffp-contract=off, so each '+', '-', '*', '/'
 * below is one IEEE-754 double operation exactly as in the reference build; log10/exp/pow
 * come from the same host libm the reference would use.
 */
#include "lod_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* garlic-roh.cpp:355-386.  Per-genotype log10 likelihood ratio autozygous / non-autozygous.
 * Genotypes outside {0,1,2} (missing is -9) and monomorphic frequencies give log10(1/1). */
double oracle_lod(int genotype, double freq, double error)
{
    double aut = 1, non = 1;
    if (freq == 0 || freq == 1) {
        /* 1 / 1 */
    } else if (genotype == 0) {
        non = (1 - freq) * (1 - freq);
        aut = (1 - error) * (1 - freq) + error * non;
    } else if (genotype == 1) {
        non = 2 * (freq) * (1 - freq);
        aut = error * non;
    } else if (genotype == 2) {
        non = (freq) * (freq);
        aut = (1 - error) * (freq) + error * non;
    }
    return log10(aut / non);
}

/* garlic-roh.cpp:11-16: closed-interval overlap of [qStart,qEnd] with [tStart,tEnd]. */
int oracle_in_gap(int qStart, int qEnd, int tStart, int tEnd)
{
    if (tStart <= qStart && tEnd >= qStart) return 1;
    if (tStart <= qEnd && tEnd >= qEnd) return 1;
    if (tStart >= qStart && tEnd <= qEnd) return 1;
    return 0;
}

/* garlic-roh.cpp:134-136 */
double oracle_nomut(double M, double mu, double interval)
{
    return exp(-2.0 * M * mu * interval);
}

/* garlic-roh.cpp:138-140 (nomut with mu = 1) */
double oracle_norec(double M, double interval)
{
    return oracle_nomut(M, 1, interval);
}

/* garlic-data.cpp:1557-1576 */
double oracle_tgls_to_error(double gl, int gl_type)
{
    if (gl_type == 0) {            /* GQ */
        gl /= (-10.0);
        gl = (gl > -10) ? gl : -10;
        gl = pow(10, gl);
    } else if (gl_type == 1) {     /* GL */
        gl = (gl > -10) ? gl : -10;
        gl = 1 - pow(10, gl);
    } else {                       /* PL */
        gl /= (-10.0);
        gl = (gl > -10) ? gl : -10;
        gl = 1 - pow(10, gl);
    }
    if (gl <= 0) gl = 0.0000000000000001;
    if (gl > 1) gl = 1;
    return gl;
}

/* A pair of consecutive SNPs (a then b) breaks a window when it is wider than max_gap or
 * touches the centromere: garlic-roh.cpp:60-61, 82-83, 109-110, 261-262. */
static int pair_breaks(const int32_t *pos, int a, int b, int max_gap, int cStart, int cEnd)
{
    return (pos[b] - pos[a] > max_gap) || oracle_in_gap(pos[a], pos[b], cStart, cEnd);
}

/* Fresh left-to-right sum of one window (garlic-roh.cpp:57-71 and 106-120).  Returns the
 * index the scan must resume from (the reference's `locus = prevI`), or -1 when the window
 * is complete. */
static int fresh_window(int locus, int winsize, int ind, int nind, const int16_t *g,
                        const double *freq, const int32_t *pos, const double *gl,
                        int cStart, int cEnd, double error, int max_gap, double *acc)
{
    int prev = locus;
    double sum = 0;
    for (int i = locus; i < locus + winsize; i++) {
        if (pair_breaks(pos, prev, i, max_gap, cStart, cEnd)) {
            *acc = ORACLE_MISSING;
            return prev;
        }
        if (gl) error = gl[(size_t)i * nind + ind];
        sum += oracle_lod(g[(size_t)i * nind + ind], freq[i], error);
        prev = i;
    }
    *acc = sum;
    return -1;
}

/* garlic-roh.cpp:18-132 for individuals [ind_begin, ind_end). */
void oracle_calc_lod_range(int nloci, int nind, const int16_t *g, const double *freq,
                           const int32_t *pos, const double *gl, int cStart, int cEnd,
                           int winsize, double error, int max_gap,
                           int ind_begin, int ind_end, double *win)
{
    /* garlic-roh.cpp:43 : windows that would overshoot the last locus are never scored */
    int stop = nloci - winsize + 1;

    for (int ind = ind_begin; ind < ind_end; ind++) {
        double *w = win + (size_t)ind * nloci;
        for (int l = 0; l < nloci; l++) w[l] = ORACLE_MISSING; /* garlic-data.cpp:1633 */

        for (int locus = 0; locus < stop; locus++) {
            /* garlic-roh.cpp:55,79: first window, or previous window MISSING *by value* */
            if (locus == 0 || w[locus - 1] == ORACLE_MISSING) {
                int resume = fresh_window(locus, winsize, ind, nind, g, freq, pos, gl,
                                          cStart, cEnd, error, max_gap, &w[locus]);
                if (resume >= 0) locus = resume; /* garlic-roh.cpp:65,114 */
                continue;
            }
            int in = locus + winsize - 1; /* SNP entering the window */
            int out = locus - 1;          /* SNP leaving it */
            if (pair_breaks(pos, in - 1, in, max_gap, cStart, cEnd)) {
                w[locus] = ORACLE_MISSING;
                locus = locus + winsize - 2; /* garlic-roh.cpp:87 */
                continue;
            }
            /* garlic-roh.cpp:92-100: (prev - leaving) + entering, two roundings */
            double e_out = gl ? gl[(size_t)out * nind + ind] : error;
            double e_in = gl ? gl[(size_t)in * nind + ind] : error;
            w[locus] = w[locus - 1] - oracle_lod(g[(size_t)out * nind + ind], freq[out], e_out)
                                   + oracle_lod(g[(size_t)in * nind + ind], freq[in], e_in);
        }
    }
}

void oracle_calc_lod(int nloci, int nind, const int16_t *g, const double *freq,
                     const int32_t *pos, const double *gl, int cStart, int cEnd,
                     int winsize, double error, int max_gap, double *win)
{
    oracle_calc_lod_range(nloci, nind, g, freq, pos, gl, cStart, cEnd, winsize, error, max_gap,
                          0, nind, win);
}

void oracle_calc_lod_mt(int nloci, int nind, const int16_t *g, const double *freq,
                        const int32_t *pos, const double *gl, int cStart, int cEnd,
                        int winsize, double error, int max_gap, int nthreads, double *win)
{
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int t = 0; t < nthreads; t++) {
        int b = (int)((int64_t)nind * t / nthreads);
        int e = (int)((int64_t)nind * (t + 1) / nthreads);
        oracle_calc_lod_range(nloci, nind, g, freq, pos, gl, cStart, cEnd, winsize, error,
                              max_gap, b, e, win);
    }
}

/* garlic-roh.cpp:204-277 for loci [start, stop) -- one "thread" of the reference. */
static void wlod_slice(int start, int stop, int nloci, int nind, const int16_t *g,
                       const double *freq, const int32_t *pos, const double *gpos,
                       const double *gl, const double *ld, int cStart, int cEnd, int winsize,
                       double error, int max_gap, double mu, int M, double *win)
{
    if (nloci - stop < winsize) stop = nloci - winsize + 1; /* garlic-roh.cpp:231 */
    int size = stop - start + winsize + 1;
    if (size < 1) return;
    double *score = (double *)malloc(sizeof(double) * (size_t)size);
    int last = (stop + winsize + 1 > nloci) ? nloci : stop + winsize + 1;

    for (int ind = 0; ind < nind; ind++) {
        /* garlic-roh.cpp:244-250: lod * nomut(physical interval) * norec(genetic interval);
         * the first locus of the chromosome uses its absolute position as the interval. */
        for (int locus = start; locus < last; locus++) {
            if (gl) error = gl[(size_t)locus * nind + ind];
            double dp = (locus > 0) ? (pos[locus] - pos[locus - 1]) : pos[locus];
            double dg = (locus > 0) ? (gpos[locus] - gpos[locus - 1]) : gpos[locus];
            score[locus - start] = oracle_lod(g[(size_t)locus * nind + ind], freq[locus], error)
                                   * oracle_nomut(M, mu, dp) * oracle_norec(M, dg);
        }
        /* garlic-roh.cpp:253-273: every window summed afresh, left to right from 0 */
        double *w = win + (size_t)ind * nloci;
        for (int locus = start; locus < stop; locus++) {
            double sum = 0;
            int prev = locus;
            for (int i = locus; i < locus + winsize; i++) {
                if (pair_breaks(pos, prev, i, max_gap, cStart, cEnd)) {
                    sum = ORACLE_MISSING;
                    w[locus] = sum;
                    locus = prev; /* garlic-roh.cpp:266 */
                    goto next_window;
                }
                sum += score[i - start] * (1.0 / ld[(size_t)locus * winsize + (i - locus)]);
                prev = i;
            }
            w[locus] = sum;
        next_window:;
        }
    }
    free(score);
}

void oracle_calc_wlod(int nloci, int nind, const int16_t *g, const double *freq,
                      const int32_t *pos, const double *gpos, const double *gl, const double *ld,
                      int cStart, int cEnd, int winsize, double error, int max_gap,
                      double mu, int M, int nthreads, double *win)
{
    for (size_t k = 0; k < (size_t)nind * nloci; k++) win[k] = ORACLE_MISSING; /* garlic-data.cpp:1633 */
    /* garlic-data.cpp:538-555 */
    if (nthreads > nloci) nthreads = nloci;
    if (nthreads < 1) nthreads = 1;
    int div = nloci / nthreads, rem = nloci % nthreads;
    int *start = (int *)malloc(sizeof(int) * (size_t)(nthreads + 1));
    start[0] = 0;
    for (int t = 0; t < nthreads; t++) start[t + 1] = start[t] + div + (t < rem ? 1 : 0);
#pragma omp parallel for num_threads(nthreads) schedule(static, 1)
    for (int t = 0; t < nthreads; t++)
        wlod_slice(start[t], start[t + 1], nloci, nind, g, freq, pos, gpos, gl, ld, cStart, cEnd,
                   winsize, error, max_gap, mu, M, win);
    free(start);
}

/* garlic-data.cpp:656-676 */
void oracle_geno_freq(int nloci, int nind, const int16_t *g, double *hom_freq)
{
    for (int locus = 0; locus < nloci; locus++) {
        double total = 0, hom = 0;
        for (int ind = 0; ind < nind; ind++) {
            int v = g[(size_t)locus * nind + ind];
            if (v != -9) {
                if (v == 2 || v == 0) hom++;
                total++;
            }
        }
        hom /= total;
        hom_freq[locus] = hom;
    }
}

/* garlic-data.cpp:558-583 */
static double hr2_pair(int nind, const int16_t *g, const double *hom_freq, int i, int j,
                       const int32_t *idx, int n_idx)
{
    double HA = hom_freq[i], HB = hom_freq[j];
    if (!(HA > 0 && HA < 1 && HB > 0 && HB < 1)) return 0;
    double HAB = 0, total = 0;
    for (int k = 0; k < n_idx; k++) {
        int ind = idx[k];
        int a = g[(size_t)i * nind + ind], b = g[(size_t)j * nind + ind];
        if (a != -9 && b != -9) {
            total++;
            if (a != 1 && b != 1) HAB++;
        }
    }
    HAB /= total;
    double H = HAB - HA * HB;
    double v = H * H / (HA * (1 - HA) * HB * (1 - HB));
    return (v > 1) ? 1 : v;
}

/* garlic-data.cpp:474-527: LD[s][k] = sum over i in window s (in order) of hr2(i, s+k),
 * with the i == s+k term contributing exactly 1. */
void oracle_hr2_ld(int nloci, int nind, const int16_t *g, const double *hom_freq, int winsize,
                   const int32_t *idx, int n_idx, double *ld)
{
    for (size_t k = 0; k < (size_t)nloci * winsize; k++) ld[k] = 0; /* garlic-data.cpp:619-631 */
    int stop = nloci - winsize + 1;
#pragma omp parallel for schedule(dynamic, 64)
    for (int s = 0; s < stop; s++) {
        for (int site = s; site < s + winsize; site++) {
            double acc = 0;
            for (int i = s; i <= s + winsize - 1; i++) {
                if (i != site) acc += hr2_pair(nind, g, hom_freq, i, site, idx, n_idx);
                else acc += 1;
            }
            ld[(size_t)s * winsize + (site - s)] = acc;
        }
    }
}

/* garlic-data.cpp:585-617 (--phased): r2 from the count of chromosomes carrying the counted allele
 * at both SNPs; a double heterozygote counts once if its first copies agree. */
static double r2_pair(int nind, const int16_t *g, const uint8_t *fc, const double *freq, int i, int j,
                      const int32_t *idx, int n_idx)
{
    double pi = freq[i], pj = freq[j];
    if (!(pi > 0 && pi < 1 && pj > 0 && pj < 1)) return 0;
    double x11 = 0, total = 0;
    for (int k = 0; k < n_idx; k++) {
        int ind = idx[k];
        int a = g[(size_t)i * nind + ind], b = g[(size_t)j * nind + ind];
        if (a != -9 && b != -9) {
            total += 2;
            if (a == 2 && b == 2) x11 += 2;
            else if (a == 1 && b == 2) x11++;
            else if (a == 2 && b == 1) x11++;
            else if (a == 1 && b == 1 && fc[(size_t)j * nind + ind] == fc[(size_t)i * nind + ind]) x11++;
        }
    }
    x11 /= total;
    double D = x11 - pi * pj;
    double v = D * D / (pi * (1 - pi) * pj * (1 - pj));
    return (v > 1) ? 1 : v;
}

/* garlic-data.cpp:426-535 (calcR2LD / parallelR2 / ldR2): as oracle_hr2_ld with r2 */
void oracle_r2_ld(int nloci, int nind, const int16_t *g, const uint8_t *first_copy, const double *freq,
                  int winsize, const int32_t *idx, int n_idx, double *ld)
{
    for (size_t k = 0; k < (size_t)nloci * winsize; k++) ld[k] = 0;
    int stop = nloci - winsize + 1;
#pragma omp parallel for schedule(dynamic, 64)
    for (int s = 0; s < stop; s++) {
        for (int site = s; site < s + winsize; site++) {
            double acc = 0;
            for (int i = s; i <= s + winsize - 1; i++) {
                if (i != site) acc += r2_pair(nind, g, first_copy, freq, i, site, idx, n_idx);
                else acc += 1;
            }
            ld[(size_t)s * winsize + (site - s)] = acc;
        }
    }
}

/* garlic-data.cpp:2026-2069 */
int64_t oracle_flatten(int nloci, int nind, const double *win, int step, double *out)
{
    int64_t n = 0;
    for (int ind = 0; ind < nind; ind++)
        for (int locus = 0; locus < nloci; locus += step) {
            double x = win[(size_t)ind * nloci + locus];
            if (x != ORACLE_MISSING && !isnan(x)) out[n++] = x;
        }
    return n;
}

/* garlic-data.cpp:2071-2150 (convertSubsetWinData2DoubleData), one chromosome, with the individuals
 * the reference draws (gsl_ran_choose, time-seeded; GSL is absent from the mount, so the function
 * itself cannot be linked into oracle/_ref) supplied by the caller: the same two loops with
 * data[randInd[ind]] in place of data[ind].  Pinned against the real convertWinData2DoubleData
 * applied to the rows win[randInd] (tests/test_oracle_vs_ref.py::test_subset_flatten). */
int64_t oracle_flatten_subset(int nloci, int nind, const double *win, int step, const int32_t *rand_ind,
                              int n_sub, double *out)
{
    int64_t n = 0;
    (void)nind;
    for (int ind = 0; ind < n_sub; ind++)
        for (int locus = 0; locus < nloci; locus += step) {
            double x = win[(size_t)rand_ind[ind] * nloci + locus];
            if (x != ORACLE_MISSING && !isnan(x)) out[n++] = x;
        }
    return n;
}

/* Validity of each window start as garlic-roh.cpp:50-125 leaves it (shared by all
 * individuals): s < nloci-W+1, SNP s itself not inside the centromere, and no breaking
 * pair (k-1,k) for s < k <= s+W-1. */
void oracle_mask(int nloci, const int32_t *pos, int cStart, int cEnd, int winsize, int max_gap,
                 uint8_t *valid)
{
    int stop = nloci - winsize + 1;
    for (int s = 0; s < nloci; s++) {
        int ok = (s < stop) && !oracle_in_gap(pos[s], pos[s], cStart, cEnd);
        for (int k = s + 1; ok && k <= s + winsize - 1; k++)
            if (pair_breaks(pos, k - 1, k, max_gap, cStart, cEnd)) ok = 0;
        valid[s] = (uint8_t)ok;
    }
}

/* garlic-roh.cpp:446-454: coverage of every SNP by above-cutoff windows of one individual row.
 * (The reference increments inWin[w + i] without a bound; a qualifying window needs a real score,
 * and real scores end at nloci - winsize, so the bound below only matters for cutoff <= MISSING.) */
void oracle_roh_coverage(int nloci, int nind, const double *win, int winsize, double cutoff,
                         int16_t *inwin)
{
    for (int ind = 0; ind < nind; ind++) {
        int16_t *in = inwin + (size_t)ind * nloci;
        const double *row = win + (size_t)ind * nloci;
        for (int w = 0; w < nloci; w++) in[w] = 0;
        for (int w = 0; w < nloci; w++)
            if (row[w] >= cutoff)
                for (int i = 0; i < winsize && w + i < nloci; i++) in[w + i]++;
    }
}

/* garlic-roh.cpp:456-533: the second half of assembleROHWindows for one individual row of coverage counts --
 * SNPs whose count reaches OVERLAP_FRAC * winsize (clamped to [1, winsize], :421-423) are strung into
 * segments, split where two neighbouring SNPs are more than max_gap apart or straddle the centromere,
 * closed at the last SNP; a segment is reported when its SNP count reaches the same threshold.  The four
 * branches in the reference's order, tests on winStart (a position) as written there: "< 0" for "none",
 * "> 0" for "one is open".  Returns the number of segments; (start, stop) SNP indices, at most cap of them. */
int oracle_roh_segments(int nloci, const int16_t *inwin, const int32_t *pos, int cStart, int cEnd, int winsize,
                        int max_gap, double overlap_frac, int cap, int32_t *seg_start, int32_t *seg_stop)
{
    double thr = overlap_frac * winsize;
    thr = (thr >= 1) ? thr : 1;
    thr = (thr <= winsize) ? thr : winsize;
    int n = 0;
    int winStart = -1, winStartIndex = -1;
    for (int w = 0; w < nloci; w++) {
        int stopIndex = -2;
        if (winStart < 0 && inwin[w] >= thr) {
            winStart = pos[w];
            winStartIndex = w;
            continue;
        } else if (inwin[w] >= thr && pair_breaks(pos, w - 1, w, max_gap, cStart, cEnd)) {
            stopIndex = w - 1;
        } else if (winStart > 0 && !(inwin[w] >= thr)) {
            stopIndex = w - 1;
        } else if (winStart > 0 && w + 1 >= nloci) {
            stopIndex = w;
        }
        if (stopIndex == -2) continue;
        if (stopIndex - winStartIndex + 1 >= thr) {
            if (n < cap) { seg_start[n] = winStartIndex; seg_stop[n] = stopIndex; }
            n++;
        }
        if (inwin[w] >= thr && stopIndex == w - 1) {     /* the second branch: the next segment starts here */
            winStart = pos[w];
            winStartIndex = w;
        } else {
            winStart = -1;
            winStartIndex = -1;
        }
    }
    return n;
}
