/* TEST INFRASTRUCTURE ONLY -- CPU restatement of GARLIC's Phase-I LOD path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this
 * library, and only as the checker / reported baseline.  The product path
 * (garlic_amd/csrc, libgarlic_hip.so) never links or calls it.
 *
 * Parity status: PINNED.  The reference has no tests or golden vectors for this path
 * (SURVEY.md section 4), so every function here is checked bit-for-bit against the real
 * reference sources compiled into oracle/_ref/libgarlic_ref.so (tests/test_oracle_vs_ref.py,
 * build container only) and against fixtures generated from that build and committed
 * under tests/golden/ (tools/make_golden.py).
 * oracle_roh_coverage restates six lines in the middle of assembleROHWindows (garlic-roh.cpp:446-454),
 * which the reference does not expose as a function; it is pinned through what those counts feed:
 * tests/test_oracle_vs_ref.py derives the ROH segments from them (second half of that function,
 * restated in the test) and compares with the segments the real assembleROHWindows reports, for
 * overlap thresholds from one SNP to the whole window; a brute-force numpy loop checks it everywhere
 * else (tests/test_oracle_golden.py).
 *
 * All arrays are flat, row-major:
 *   genotypes  int16  [nloci][nind]   (reference HapData::data, garlic-data.h:35)
 *   gl         double [nloci][nind]   (reference GenoLikeData::data, garlic-data.h:91) or NULL
 *   win        double [nind][nloci]   (reference WinData::data, garlic-data.h:83)
 *   ld         double [nloci][winsize](reference LDData::LD, garlic-data.h:105)
 */
#ifndef LOD_ORACLE_H
#define LOD_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_MISSING (-9999.0) /* garlic-data.h:24 */

double oracle_lod(int genotype, double freq, double error);           /* garlic-roh.cpp:355-386 */
int    oracle_in_gap(int qStart, int qEnd, int tStart, int tEnd);      /* garlic-roh.cpp:11-16 */
double oracle_nomut(double M, double mu, double interval);             /* garlic-roh.cpp:134-136 */
double oracle_norec(double M, double interval);                        /* garlic-roh.cpp:138-140 */

/* garlic-data.cpp:1557-1576; gl_type: 0=GQ 1=GL 2=PL */
double oracle_tgls_to_error(double value, int gl_type);

/* garlic-roh.cpp:18-132 (one chromosome); win is fully written (prefill -9999 as
 * initWinData does, garlic-data.cpp:1633).  Individuals [ind_begin, ind_end) only. */
void oracle_calc_lod_range(int nloci, int nind, const int16_t *genotypes, const double *freq,
                           const int32_t *pos, const double *gl, int cStart, int cEnd,
                           int winsize, double error, int max_gap,
                           int ind_begin, int ind_end, double *win);

void oracle_calc_lod(int nloci, int nind, const int16_t *genotypes, const double *freq,
                     const int32_t *pos, const double *gl, int cStart, int cEnd,
                     int winsize, double error, int max_gap, double *win);

/* Same result, individuals split over nthreads OpenMP threads (the reference itself is
 * single-threaded here; this is the "all host cores" baseline of SURVEY.md 8(d)). */
void oracle_calc_lod_mt(int nloci, int nind, const int16_t *genotypes, const double *freq,
                        const int32_t *pos, const double *gl, int cStart, int cEnd,
                        int winsize, double error, int max_gap, int nthreads, double *win);

/* garlic-roh.cpp:144-277 (calcwLOD + parallelwLOD), loci split over nthreads as
 * make_thread_partition does (garlic-data.cpp:538-555). */
void oracle_calc_wlod(int nloci, int nind, const int16_t *genotypes, const double *freq,
                      const int32_t *pos, const double *gpos, const double *gl, const double *ld,
                      int cStart, int cEnd, int winsize, double error, int max_gap,
                      double mu, int M, int nthreads, double *win);

/* garlic-data.cpp:656-676 */
void oracle_geno_freq(int nloci, int nind, const int16_t *genotypes, double *hom_freq);

/* garlic-data.cpp:377-583 (calcHR2LD / parallelHR2 / ldHR2 / hr2), explicit subset index. */
void oracle_hr2_ld(int nloci, int nind, const int16_t *genotypes, const double *hom_freq,
                   int winsize, const int32_t *ind_index, int n_index, double *ld);

/* garlic-data.cpp:426-535, 585-617 (calcR2LD / r2, --phased); first_copy = HapData::firstCopy,
 * uint8 [nloci][nind]; freq = FreqData::freq. */
void oracle_r2_ld(int nloci, int nind, const int16_t *genotypes, const uint8_t *first_copy,
                  const double *freq, int winsize, const int32_t *ind_index, int n_index, double *ld);

/* garlic-data.cpp:2026-2069, one chromosome; returns number of values written. */
int64_t oracle_flatten(int nloci, int nind, const double *win, int step, double *out);
/* garlic-data.cpp:2071-2150 with the drawn individuals rand_ind[0 .. n_sub) supplied */
int64_t oracle_flatten_subset(int nloci, int nind, const double *win, int step, const int32_t *rand_ind,
                              int n_sub, double *out);
void oracle_roh_coverage(int nloci, int nind, const double *win, int winsize, double cutoff,
                         int16_t *inwin);

/* Window validity mask implied by garlic-roh.cpp:50-125 (SURVEY.md 8(a') item 1);
 * valid[s]=1 iff window s holds a LOD score for every individual. */
void oracle_mask(int nloci, const int32_t *pos, int cStart, int cEnd, int winsize, int max_gap,
                 uint8_t *valid);

#ifdef __cplusplus
}
#endif
#endif
