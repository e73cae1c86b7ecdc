// TEST INFRASTRUCTURE ONLY -- never linked, imported or executed by the product path.
//
// The reference-side binding of INTEGRATION.md, compiled against the reference's OWN headers
// (/root/reference/src/garlic-roh.h:96-112, garlic-data.h:32-108, where they lie) and linked with the
// reference's own objects (oracle/Makefile -> oracle/_ref/libgarlic_ref_hip.so) and libgarlic_hip.so:
//   * hip_calcLODWindows / hip_calcwLODWindows are the bodies a GARLIC maintainer would give calcLODWindows
//     (src/garlic-roh.cpp:279-309) and calcwLODWindows (:311-347) -- they take the reference's
//     vector<HapData*>* .. centromere* and return a vector<WinData*>* made by the reference's initWinData, which the
//     reference's releaseWinData (src/garlic-data.cpp:1640-1667) frees.  INTEGRATION.md's code block is generated
//     from the marked region below (tools/gen_integration.py), so the two cannot drift;
//   * hip_assembleROHWindows is what main would call for calcLODWindows + assembleROHWindows (:409-545) when the raw
//     scores are not asked for: the ROH segments straight from the device (garlic_roh_segments);
//   * refbind_compare_* build the reference's structs from flat arrays, run the reference's calcLODWindows /
//     calcwLODWindows (/ assembleROHWindows on its scores) and the binding on the SAME objects and memcmp the WinData
//     rows (the ROHData lists).
#include "garlic-roh.h"
#include "garlic-data.h"
#include "garlic-centromeres.h"
#include "garlic_hip.h"                     // include/garlic_hip.h of this repository

#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <unistd.h>
#include <vector>

// INTEGRATION-BEGIN
// garlic-roh.cpp -- the bodies of calcLODWindows (lines 279-309) and calcwLODWindows (lines 311-347) become:
namespace {
struct HipPanel {      // context + device-resident panel of one call, released on every way out
    garlic_ctx *ctx = NULL;
    garlic_panel *panel = NULL;
    ~HipPanel() { if (panel) garlic_panel_destroy(panel); if (ctx) garlic_ctx_destroy(ctx); }
};

// the reference's per-chromosome structs -> one resident panel (small per-SNP arrays flattened, genotype rows uploaded
// one SNP at a time: HapData rows are separate allocations)
void hip_upload(HipPanel &h, vector< HapData * > *hap, vector< FreqData * > *frq, vector< MapData * > *map,
                vector< GenoLikeData * > *gl, centromere *centro, bool USE_GL, bool withGeneticPos)
{
    const int nchr = (int)map->size(), nind = hap->at(0)->nind;
    vector<int32_t> chr_nloci, cs, ce, pos;
    vector<double> freq, gpos;
    for (int c = 0; c < nchr; c++) {
        MapData *m = map->at(c);
        chr_nloci.push_back(m->nloci);
        cs.push_back(centro->centromereStart(m->chr));        // garlic-roh.cpp:36-37
        ce.push_back(centro->centromereEnd(m->chr));
        pos.insert(pos.end(), m->physicalPos, m->physicalPos + m->nloci);
        if (withGeneticPos) gpos.insert(gpos.end(), m->geneticPos, m->geneticPos + m->nloci);
        freq.insert(freq.end(), frq->at(c)->freq, frq->at(c)->freq + m->nloci);
    }
    if (garlic_ctx_create(0, NULL, &h.ctx) || garlic_panel_create(h.ctx, nchr, chr_nloci.data(), nind, &h.panel)) throw 0;
    if (garlic_panel_set_map(h.panel, pos.data(), withGeneticPos ? gpos.data() : NULL, cs.data(), ce.data())) throw 0;
    if (garlic_panel_set_freq(h.panel, freq.data())) throw 0;
    int64_t g = 0;
    for (int c = 0; c < nchr; c++)
        for (int l = 0; l < chr_nloci[c]; l++, g++) {
            if (garlic_panel_set_genotypes(h.panel, hap->at(c)->data[l], nind, g, 1, GARLIC_HOST)) throw 0;
            // GLDataByChr may be an uninitialised pointer when !USE_GL (garlic-roh.cpp:713,720): never touched then
            if (USE_GL && garlic_panel_set_gl(h.panel, gl->at(c)->data[l], nind, g, 1, GARLIC_HOST)) throw 0;
        }
}

// scores of all chromosomes (dense rows = WinData rows) -> the reference's WinData, which the caller frees with releaseWinData
vector< WinData * > *hip_download(HipPanel &h, vector< MapData * > *map, int nind, bool weighted, int winsize, double error,
                                  int MAX_GAP, bool USE_GL, int M, double mu)
{
    const int nchr = (int)map->size();
    vector<int64_t> base(nchr), pitch(nchr);
    int64_t total = 0;
    if (garlic_lod_out_layout(h.panel, 1, nind, base.data(), pitch.data(), &total)) throw 0;
    vector<double> out((size_t)total);
    const int rc = weighted ? garlic_wlod_windows(h.panel, winsize, error, MAX_GAP, USE_GL, M, mu, 0, nind, 1, out.data(), GARLIC_HOST)
                            : garlic_lod_windows(h.panel, winsize, error, MAX_GAP, USE_GL, 0, nind, 1, out.data(), GARLIC_HOST);
    if (rc) throw 0;                                           // the reference's error convention (garlic-data.cpp:1619)
    vector< WinData * > *win = initWinData(map, nind);         // garlic-data.cpp:1690 (its -9999 prefill is overwritten)
    for (int c = 0; c < nchr; c++)
        for (int i = 0; i < nind; i++)
            memcpy(win->at(c)->data[i], &out[(size_t)(base[c] + i * pitch[c])], sizeof(double) * (size_t)map->at(c)->nloci);
    return win;
}
}

vector< WinData * > *hip_calcLODWindows(vector< HapData * > *hapDataByChr, vector< FreqData * > *freqDataByChr,
                                        vector< MapData * > *mapDataByChr, vector< GenoLikeData * > *GLDataByChr,
                                        centromere *centro, int winsize, double error, int MAX_GAP, bool USE_GL)
{
    HipPanel h;
    hip_upload(h, hapDataByChr, freqDataByChr, mapDataByChr, GLDataByChr, centro, USE_GL, false);
    return hip_download(h, mapDataByChr, hapDataByChr->at(0)->nind, false, winsize, error, MAX_GAP, USE_GL, 0, 0.0);
}

vector< WinData * > *hip_calcwLODWindows(vector< HapData * > *hapDataByChr, vector< FreqData * > *freqDataByChr,
                                         vector< MapData * > *mapDataByChr, vector< GenoLikeData * > *GLDataByChr,
                                         vector< LDData * > *ldDataByChr, centromere *centro, int winsize, double error,
                                         int MAX_GAP, bool USE_GL, int M, double mu, int numThreads)
{
    (void)numThreads;                                          // the reference's result does not depend on it either
    HipPanel h;
    hip_upload(h, hapDataByChr, freqDataByChr, mapDataByChr, GLDataByChr, centro, USE_GL, true);
    // the LDData calcLDData made (garlic-data.cpp:330-375), flattened [locus][winsize]; garlic_panel_compute_ld would
    // compute the same weights on the device and nothing LD-sized would cross PCIe
    vector<double> flat;
    for (size_t c = 0; c < ldDataByChr->size(); c++)
        for (int l = 0; l < ldDataByChr->at(c)->nloci; l++)
            flat.insert(flat.end(), ldDataByChr->at(c)->LD[l], ldDataByChr->at(c)->LD[l] + winsize);
    if (garlic_panel_set_ld(h.panel, winsize, flat.data(), GARLIC_HOST)) throw 0;
    return hip_download(h, mapDataByChr, hapDataByChr->at(0)->nind, true, winsize, error, MAX_GAP, USE_GL, M, mu);
}

// garlic-main.cpp:346-420 -- when --raw-lod is not asked for, calcLODWindows + assembleROHWindows (garlic-roh.cpp:409-545)
// become one call: the device goes from the genotypes to the ROH segments, neither the window scores nor the per-SNP
// coverage counts exist anywhere.  Same results as assembleROHWindows(calcLODWindows(..), ..): rohData per individual,
// the pooled lengths in the order the reference appends them.
vector< ROHData * > *hip_assembleROHWindows(vector< HapData * > *hapDataByChr, vector< FreqData * > *freqDataByChr,
                                            vector< MapData * > *mapDataByChr, vector< GenoLikeData * > *GLDataByChr,
                                            IndData *indData, centromere *centro, double lodScoreCutoff, ROHLength **rohLength,
                                            int winSize, double error, int MAX_GAP, double OVERLAP_FRAC, bool CM, bool USE_GL)
{
    HipPanel h;
    hip_upload(h, hapDataByChr, freqDataByChr, mapDataByChr, GLDataByChr, centro, USE_GL, false);
    int64_t n = 0;
    vector<garlic_roh_segment> seg((size_t)(64 * indData->nind));      // room for a first guess; the call says how many there are
    for (int attempt = 0; attempt < 2; attempt++) {
        if (garlic_roh_segments(h.panel, winSize, error, MAX_GAP, USE_GL, 0, 0, 0.0, lodScoreCutoff, OVERLAP_FRAC, seg.data(),
                                (int64_t)seg.size(), &n)) throw 0;
        if (n <= (int64_t)seg.size()) break;
        seg.resize((size_t)n);                                           // (nothing usable was written: once more, with room)
    }
    vector< ROHData * > *rohDataByInd = initROHData(indData);           // garlic-roh.cpp:387
    for (int ind = 0; ind < indData->nind; ind++) rohDataByInd->at(ind)->indID = indData->indID[ind];
    (*rohLength) = initROHLength((int)n, indData->pop);                 // :535-541
    for (int64_t k = 0; k < n; k++) {                                    // ordered by individual, chromosome, position
        MapData *mapData = mapDataByChr->at(seg[k].chr);
        ROHData *rohData = rohDataByInd->at(seg[k].ind);
        const int winStart = mapData->physicalPos[seg[k].start], winStop = mapData->physicalPos[seg[k].stop];
        const double size = CM ? mapData->geneticPos[seg[k].stop] - mapData->geneticPos[seg[k].start] : winStop - winStart + 1;
        (*rohLength)->length[k] = size;
        rohData->length.push_back(size);
        rohData->chr.push_back(seg[k].chr);
        rohData->start.push_back(winStart);
        rohData->stop.push_back(winStop);
    }
    return rohDataByInd;
}
// INTEGRATION-END

#define REF_API extern "C" __attribute__((visibility("default")))

namespace {

struct StderrSilencer {      // the reference prints a progress bar per chromosome
    int saved;
    StderrSilencer() { fflush(stderr); saved = dup(2); int n = open("/dev/null", O_WRONLY); if (n >= 0) { dup2(n, 2); close(n); } }
    ~StderrSilencer() { std::cerr.flush(); fflush(stderr); if (saved >= 0) { dup2(saved, 2); close(saved); } }
};

// the reference's structs from flat arrays: chromosome c holds loci chr_off[c] .. chr_off[c+1]
struct RefData {
    vector< HapData * > *hap = new vector< HapData * >;
    vector< FreqData * > *frq = new vector< FreqData * >;
    vector< MapData * > *map = new vector< MapData * >;
    vector< GenoLikeData * > *gl = NULL;
    centromere *centro = new centromere();
    ~RefData()
    {
        releaseHapData(hap);
        releaseFreqData(frq);
        releaseMapData(map);
        if (gl) releaseGLData(gl);
        delete centro;
    }
};

void build(RefData &d, int nchr, const int *chr_nloci, int nind, const short *geno, const double *freq, const int *pos,
           const double *gpos, const int *cStart, const int *cEnd, const double *gl)
{
    char path[] = "/tmp/garlic_refbind_centro_XXXXXX";
    int fd = mkstemp(path);
    FILE *cf = fd >= 0 ? fdopen(fd, "w") : NULL;
    if (gl) d.gl = new vector< GenoLikeData * >;
    int64_t off = 0;
    for (int c = 0; c < nchr; c++) {
        const int n = chr_nloci[c];
        HapData *h = initHapData((unsigned)nind, (unsigned)n, false);
        FreqData *f = initFreqData(n);
        MapData *m = initMapData(n);
        m->chr = "chrB" + std::to_string(c + 1);
        GenoLikeData *g = gl ? initGLData((unsigned)nind, (unsigned)n) : NULL;
        for (int l = 0; l < n; l++) {
            for (int i = 0; i < nind; i++) h->data[l][i] = geno[(off + l) * nind + i];
            f->freq[l] = freq[off + l];
            m->physicalPos[l] = pos[off + l];
            m->geneticPos[l] = gpos ? gpos[off + l] : 0.0;
            if (g) for (int i = 0; i < nind; i++) g->data[l][i] = gl[(off + l) * nind + i];
        }
        d.hap->push_back(h);
        d.frq->push_back(f);
        d.map->push_back(m);
        if (g) d.gl->push_back(g);
        if (cf && cStart[c] >= 0) fprintf(cf, "%s %d %d\n", m->chr.c_str(), cStart[c], cEnd[c]);   // < 0: unknown chromosome (0, 0)
        off += n;
    }
    if (cf) {
        fclose(cf);
        d.centro->readCustomCentromeres(path);
        unlink(path);
    }
}

// rows that differ between two results of the same shape; both are freed with the reference's releaseWinData
int64_t compare_and_release(vector< WinData * > *a, vector< WinData * > *b)
{
    int64_t bad = (a->size() == b->size()) ? 0 : 1;
    for (size_t c = 0; c < a->size() && c < b->size(); c++) {
        if (a->at(c)->nind != b->at(c)->nind || a->at(c)->nloci != b->at(c)->nloci) { bad++; continue; }
        for (int i = 0; i < a->at(c)->nind; i++)
            bad += memcmp(a->at(c)->data[i], b->at(c)->data[i], sizeof(double) * (size_t)a->at(c)->nloci) != 0;
    }
    releaseWinData(a);
    releaseWinData(b);
    return bad;
}

} // namespace

// genotypes [nloci][nind] (all chromosomes, SNP-major), gl likewise or NULL; cStart[c] < 0: chromosome unknown to the
// centromere table.  Returns the number of WinData rows that differ between the reference and the binding (0 =
// identical), -1 if either threw.
REF_API long refbind_compare_lod(int nchr, const int *chr_nloci, int nind, const short *geno, const double *freq,
                                 const int *pos, const int *cStart, const int *cEnd, const double *gl, int winsize,
                                 double error, int max_gap)
{
    StderrSilencer quiet;
    try {
        RefData d;
        build(d, nchr, chr_nloci, nind, geno, freq, pos, NULL, cStart, cEnd, gl);
        vector< WinData * > *ref = calcLODWindows(d.hap, d.frq, d.map, d.gl, d.centro, winsize, error, max_gap, gl != NULL);
        vector< WinData * > *mine = hip_calcLODWindows(d.hap, d.frq, d.map, d.gl, d.centro, winsize, error, max_gap, gl != NULL);
        return (long)compare_and_release(ref, mine);
    } catch (...) {
        return -1;
    }
}

// the same for calcwLODWindows; the LD weights are the reference's calcHR2LD over ALL individuals (calcLDData draws
// its subsample with GSL, which the mount lacks; the explicit index is what the other harness uses too)
REF_API long refbind_compare_wlod(int nchr, const int *chr_nloci, int nind, const short *geno, const double *freq,
                                  const int *pos, const double *gpos, const int *cStart, const int *cEnd, const double *gl,
                                  int winsize, double error, int max_gap, int M, double mu, int numThreads)
{
    StderrSilencer quiet;
    try {
        RefData d;
        build(d, nchr, chr_nloci, nind, geno, freq, pos, gpos, cStart, cEnd, gl);
        vector< LDData * > *ld = new vector< LDData * >;
        vector<int> all(nind);
        for (int i = 0; i < nind; i++) all[i] = i;
        for (int c = 0; c < nchr; c++) {
            GenoFreqData *gf = calculateGenoFreq(d.hap->at(c));
            ld->push_back(calcHR2LD(d.hap->at(c), gf, winsize, numThreads, all.data(), nind));
            releaseGenoFreq(gf);
        }
        vector< WinData * > *ref = calcwLODWindows(d.hap, d.frq, d.map, d.gl, ld, d.centro, winsize, error, max_gap, gl != NULL,
                                                   M, mu, numThreads);
        vector< WinData * > *mine = hip_calcwLODWindows(d.hap, d.frq, d.map, d.gl, ld, d.centro, winsize, error, max_gap,
                                                        gl != NULL, M, mu, numThreads);
        releaseLDData(ld);
        return (long)compare_and_release(ref, mine);
    } catch (...) {
        return -1;
    }
}

// calcLODWindows + assembleROHWindows of the reference against hip_assembleROHWindows on the same structs: the number of
// differences over every individual's chr / start / stop / length lists and the pooled length list (0 = identical;
// *n_segments: how many segments the reference reported), -1 if either threw
REF_API long refbind_compare_roh(int nchr, const int *chr_nloci, int nind, const short *geno, const double *freq,
                                 const int *pos, const double *gpos, const int *cStart, const int *cEnd, const double *gl,
                                 int winsize, double error, int max_gap, double cutoff, double overlap_frac, int cm,
                                 long *n_segments)
{
    StderrSilencer quiet;
    try {
        RefData d;
        build(d, nchr, chr_nloci, nind, geno, freq, pos, gpos, cStart, cEnd, gl);
        IndData ind;
        ind.pop = "POP";
        ind.nind = nind;
        ind.indID = new string[nind];
        for (int i = 0; i < nind; i++) ind.indID[i] = "ind" + std::to_string(i);
        vector< WinData * > *win = calcLODWindows(d.hap, d.frq, d.map, d.gl, d.centro, winsize, error, max_gap, gl != NULL);
        ROHLength *lenRef = NULL, *lenMine = NULL;
        vector< ROHData * > *ref = assembleROHWindows(win, d.map, &ind, d.centro, cutoff, &lenRef, winsize, max_gap, overlap_frac, cm != 0);
        releaseWinData(win);
        vector< ROHData * > *mine = hip_assembleROHWindows(d.hap, d.frq, d.map, d.gl, &ind, d.centro, cutoff, &lenMine, winsize,
                                                          error, max_gap, overlap_frac, cm != 0, gl != NULL);
        long bad = (lenRef->size == lenMine->size) ? 0 : 1;
        for (int k = 0; k < (int)lenRef->size && k < (int)lenMine->size; k++) bad += memcmp(&lenRef->length[k], &lenMine->length[k], sizeof(double)) != 0;
        bad += lenRef->pop != lenMine->pop;
        for (int i = 0; i < nind; i++) {
            ROHData *a = ref->at(i), *b = mine->at(i);
            bad += a->indID != b->indID;
            bad += a->chr != b->chr;
            bad += a->start != b->start;
            bad += a->stop != b->stop;
            bad += a->length.size() != b->length.size() ||
                   (a->length.size() && memcmp(a->length.data(), b->length.data(), sizeof(double) * a->length.size()) != 0);
        }
        if (n_segments) *n_segments = (long)lenRef->size;
        releaseROHData(ref);
        releaseROHData(mine);
        releaseROHLength(lenRef);
        releaseROHLength(lenMine);
        delete [] ind.indID;
        return bad;
    } catch (...) {
        return -1;
    }
}
