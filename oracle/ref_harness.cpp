// TEST INFRASTRUCTURE ONLY -- never linked, imported or executed by the product path.
//
// C-ABI harness around the *real* reference implementation (szpiech/garlic v1.1.6a).
// This file is our own code: it #includes the reference headers where they lie
// (/root/reference/src) and is linked, by oracle/Makefile, against objects compiled
// from the reference's own .cpp files into oracle/_ref/libgarlic_ref.so.  No reference
// source is copied into the repository and no stand-in is written for anything the
// image lacks: functions that need the (absent) GSL archive are discarded by the
// linker (--gc-sections) because nothing reachable from the entry points below uses
// them; the link is checked with -z defs.
//
// Entry points flatten the reference's pointer-of-pointer structs
// (garlic-data.h:32-108) into plain arrays so ctypes / C callers can drive:
//   lod            garlic-roh.cpp:355
//   calcLOD        garlic-roh.cpp:18
//   calcwLOD       garlic-roh.cpp:144   (+ parallelwLOD :204)
//   calcHR2LD      garlic-data.cpp:377  (explicit individual index, no RNG)
//   calcR2LD       garlic-data.cpp:426  (--phased; same)
//   assembleROHWindows garlic-roh.cpp:409 (pins the coverage counts through the ROH segments they give)
//   calculateGenoFreq garlic-data.cpp:656
//   readTGLSData   garlic-data.cpp:1516 (GQ/GL/PL -> error probability)
//   convertWinData2DoubleData garlic-data.cpp:2026
#include "garlic-roh.h"
#include "garlic-data.h"
#include "garlic-centromeres.h"

#include <cstdio>
#include <cstring>
#include <unistd.h>
#include <fcntl.h>
#include <string>

#define REF_API extern "C" __attribute__((visibility("default")))

namespace {

// The reference prints a progress bar on cerr for every chromosome; keep test logs clean.
struct StderrSilencer {
    int saved;
    StderrSilencer() {
        fflush(stderr);
        saved = dup(2);
        int devnull = open("/dev/null", O_WRONLY);
        if (devnull >= 0) { dup2(devnull, 2); close(devnull); }
    }
    ~StderrSilencer() {
        std::cerr.flush();
        fflush(stderr);
        if (saved >= 0) { dup2(saved, 2); close(saved); }
    }
};

const char *kChr = "chrT";

// centromere has no setter: write a one-row custom centromere file and let the
// reference's own reader (garlic-centromeres.cpp:64) parse it.
centromere *make_centromere(int cStart, int cEnd, bool known)
{
    centromere *c = new centromere();
    if (!known) return c; // unknown chr => centromereStart/End return 0 (garlic-centromeres.cpp:33-59)
    char path[] = "/tmp/garlic_ref_centro_XXXXXX";
    int fd = mkstemp(path);
    if (fd < 0) return c;
    FILE *f = fdopen(fd, "w");
    fprintf(f, "%s %d %d\n", kChr, cStart, cEnd);
    fclose(f);
    c->readCustomCentromeres(path);
    unlink(path);
    return c;
}

HapData *make_hap(int nloci, int nind, const short *g)
{
    HapData *h = new HapData;
    h->nloci = nloci;
    h->nind = nind;
    h->firstCopy = NULL;
    h->data = new short*[nloci];
    for (int l = 0; l < nloci; l++) {
        h->data[l] = new short[nind];
        memcpy(h->data[l], g + (size_t)l * nind, sizeof(short) * nind);
    }
    return h;
}
void free_hap(HapData *h)
{
    for (int l = 0; l < h->nloci; l++) delete [] h->data[l];
    delete [] h->data;
    delete h;
}

MapData *make_map(int nloci, const int *pos, const double *gpos)
{
    MapData *m = new MapData;
    m->nloci = nloci;
    m->chr = kChr;
    m->locusName = NULL;
    m->allele = NULL;
    m->physicalPos = new int[nloci];
    m->geneticPos = new double[nloci];
    for (int l = 0; l < nloci; l++) {
        m->physicalPos[l] = pos[l];
        m->geneticPos[l] = gpos ? gpos[l] : 0.0;
    }
    return m;
}
void free_map(MapData *m)
{
    delete [] m->physicalPos;
    delete [] m->geneticPos;
    delete m;
}

GenoLikeData *make_gl(int nloci, int nind, const double *gl)
{
    if (!gl) return NULL;
    GenoLikeData *d = new GenoLikeData;
    d->nloci = nloci;
    d->nind = nind;
    d->data = new double*[nloci];
    for (int l = 0; l < nloci; l++) {
        d->data[l] = new double[nind];
        memcpy(d->data[l], gl + (size_t)l * nind, sizeof(double) * nind);
    }
    return d;
}
void free_gl(GenoLikeData *d)
{
    if (!d) return;
    for (int l = 0; l < d->nloci; l++) delete [] d->data[l];
    delete [] d->data;
    delete d;
}

void copy_out(WinData *w, double *out)
{
    for (int i = 0; i < w->nind; i++)
        memcpy(out + (size_t)i * w->nloci, w->data[i], sizeof(double) * w->nloci);
}

} // namespace

REF_API double ref_lod(short genotype, double freq, double error)
{
    return lod(genotype, freq, error);
}

REF_API double ref_nomut(double M, double mu, double interval) { return nomut(M, mu, interval); }
REF_API double ref_norec(double M, double interval) { return norec(M, interval); }

REF_API int ref_inGap(int qs, int qe, int ts, int te) { return inGap(qs, qe, ts, te) ? 1 : 0; }

// genotypes: short[nloci][nind]; gl: double[nloci][nind] or NULL; win_out: double[nind][nloci]
// centro_known=0 reproduces an unknown chromosome (centromere 0,0).
REF_API int ref_calcLOD(int nloci, int nind, const short *genotypes, const double *freq,
                        const int *pos, const double *gl, int cStart, int cEnd, int centro_known,
                        int winsize, double error, int max_gap, double *win_out)
{
    StderrSilencer quiet;
    try {
        HapData *hap = make_hap(nloci, nind, genotypes);
        MapData *map = make_map(nloci, pos, NULL);
        FreqData fd; fd.freq = const_cast<double *>(freq); fd.nloci = nloci;
        GenoLikeData *gld = make_gl(nloci, nind, gl);
        centromere *c = make_centromere(cStart, cEnd, centro_known != 0);
        WinData *w = initWinData((unsigned)nind, (unsigned)nloci);
        calcLOD(map, hap, &fd, gld, w, c, winsize, error, max_gap, gl != NULL);
        copy_out(w, win_out);
        releaseWinData(w);
        delete c;
        free_gl(gld);
        free_map(map);
        free_hap(hap);
    } catch (...) { return 1; }
    return 0;
}

// ld: double[nloci][winsize]; gpos: double[nloci]
REF_API int ref_calcwLOD(int nloci, int nind, const short *genotypes, const double *freq,
                         const int *pos, const double *gpos, const double *gl, const double *ld,
                         int cStart, int cEnd, int centro_known,
                         int winsize, double error, int max_gap, double mu, int M, int numThreads,
                         double *win_out)
{
    StderrSilencer quiet;
    try {
        HapData *hap = make_hap(nloci, nind, genotypes);
        MapData *map = make_map(nloci, pos, gpos);
        FreqData fd; fd.freq = const_cast<double *>(freq); fd.nloci = nloci;
        GenoLikeData *gld = make_gl(nloci, nind, gl);
        LDData *L = initLDData(nloci, winsize);
        for (int l = 0; l < nloci; l++)
            memcpy(L->LD[l], ld + (size_t)l * winsize, sizeof(double) * winsize);
        centromere *c = make_centromere(cStart, cEnd, centro_known != 0);
        WinData *w = initWinData((unsigned)nind, (unsigned)nloci);
        calcwLOD(map, hap, &fd, gld, L, w, c, winsize, error, max_gap, gl != NULL, mu, M, numThreads);
        copy_out(w, win_out);
        releaseWinData(w);
        delete c;
        releaseLDData(L);
        free_gl(gld);
        free_map(map);
        free_hap(hap);
    } catch (...) { return 1; }
    return 0;
}

// hom_out: double[nloci] (calculateGenoFreq); ld_out: double[nloci][winsize]
// ind_index: explicit individual subset (bypasses the reference's time-seeded RNG, garlic-data.cpp:346)
REF_API int ref_calcHR2LD(int nloci, int nind, const short *genotypes, int winsize, int numThreads,
                          const int *ind_index, int n_index, double *hom_out, double *ld_out)
{
    StderrSilencer quiet;
    try {
        HapData *hap = make_hap(nloci, nind, genotypes);
        GenoFreqData *gf = calculateGenoFreq(hap);
        if (hom_out) memcpy(hom_out, gf->homFreq, sizeof(double) * nloci);
        LDData *L = calcHR2LD(hap, gf, winsize, numThreads, const_cast<int *>(ind_index), n_index);
        for (int l = 0; l < nloci; l++)
            memcpy(ld_out + (size_t)l * winsize, L->LD[l], sizeof(double) * winsize);
        releaseLDData(L);
        releaseGenoFreq(gf);
        free_hap(hap);
    } catch (...) { return 1; }
    return 0;
}

// calcR2LD (--phased, garlic-data.cpp:426): first_copy = HapData::firstCopy, uint8 [nloci][nind];
// freq = FreqData::freq.  ld_out: double[nloci][winsize]
REF_API int ref_calcR2LD(int nloci, int nind, const short *genotypes, const unsigned char *first_copy,
                         const double *freq, int winsize, int numThreads, const int *ind_index,
                         int n_index, double *ld_out)
{
    StderrSilencer quiet;
    try {
        HapData *hap = make_hap(nloci, nind, genotypes);
        hap->firstCopy = new bool*[nloci];
        for (int l = 0; l < nloci; l++) {
            hap->firstCopy[l] = new bool[nind];
            for (int i = 0; i < nind; i++) hap->firstCopy[l][i] = first_copy[(size_t)l * nind + i] != 0;
        }
        FreqData fd; fd.freq = const_cast<double *>(freq); fd.nloci = nloci;
        LDData *L = calcR2LD(hap, &fd, winsize, numThreads, const_cast<int *>(ind_index), n_index);
        for (int l = 0; l < nloci; l++)
            memcpy(ld_out + (size_t)l * winsize, L->LD[l], sizeof(double) * winsize);
        releaseLDData(L);
        for (int l = 0; l < nloci; l++) delete [] hap->firstCopy[l];
        delete [] hap->firstCopy;
        hap->firstCopy = NULL;
        free_hap(hap);
    } catch (...) { return 1; }
    return 0;
}

// assembleROHWindows (garlic-roh.cpp:409-545) on one chromosome: win = WinData::data [nind][nloci].
// Returns the number of ROH segments written (individual, start, stop; at most cap), or -1.
REF_API int ref_assembleROH(int nloci, int nind, const double *win, const int *pos, int cStart, int cEnd,
                            int centro_known, double cutoff, int winsize, int max_gap, double overlap_frac,
                            int cap, int *seg_ind, double *seg_start, double *seg_stop)
{
    StderrSilencer quiet;
    try {
        MapData *map = make_map(nloci, pos, NULL);
        centromere *c = make_centromere(cStart, cEnd, centro_known != 0);
        WinData *w = initWinData((unsigned)nind, (unsigned)nloci);
        for (int i = 0; i < nind; i++) memcpy(w->data[i], win + (size_t)i * nloci, sizeof(double) * nloci);
        IndData ind;
        ind.pop = "POP";
        ind.nind = nind;
        ind.indID = new string[nind];
        for (int i = 0; i < nind; i++) ind.indID[i] = "i";
        vector<WinData *> wins; wins.push_back(w);
        vector<MapData *> maps; maps.push_back(map);
        ROHLength *len = NULL;
        vector<ROHData *> *roh = assembleROHWindows(&wins, &maps, &ind, c, cutoff, &len, winsize, max_gap,
                                                    overlap_frac, false);
        int n = 0;
        for (int i = 0; i < nind; i++)
            for (size_t k = 0; k < roh->at(i)->start.size(); k++, n++)
                if (n < cap) { seg_ind[n] = i; seg_start[n] = roh->at(i)->start[k]; seg_stop[n] = roh->at(i)->stop[k]; }
        releaseROHData(roh);
        releaseROHLength(len);
        delete [] ind.indID;
        releaseWinData(w);
        delete c;
        free_map(map);
        return n;
    } catch (...) { return -1; }
}

// loadMapScaffold + interpolateGeneticmap (garlic-data.cpp:702-757, 759-...) for ONE chromosome of a map
// file: the genetic positions the reference assigns to physical positions pos[0..n) (all inside the
// scaffold's span).  Returns the number of interpolated sites, -1 on error.
REF_API int ref_interpolate(const char *mapfile, int cStart, int cEnd, int centro_known, int nloci, const int *pos,
                            double *gpos_out)
{
    StderrSilencer quiet;
    try {
        centromere *c = make_centromere(cStart, cEnd, centro_known != 0);
        vector<GenMapScaffold *> *sc = loadMapScaffold(mapfile, c);
        MapData *map = make_map(nloci, pos, NULL);
        int n = interpolateGeneticmap(map, sc->at(0));
        memcpy(gpos_out, map->geneticPos, sizeof(double) * nloci);
        free_map(map);
        releaseGenMapScaffold(sc);
        delete c;
        return n;
    } catch (...) { return -1; }
}

// Runs the reference's TGLS reader on a text file we are given; returns error probabilities
// double[nloci][nind].  gl_type is "GQ", "GL" or "PL".
REF_API int ref_readTGLS(const char *path, int nloci, int nind, const char *gl_type, double *out)
{
    StderrSilencer quiet;
    try {
        int *pos = new int[nloci];
        for (int l = 0; l < nloci; l++) pos[l] = l + 1;
        MapData *map = make_map(nloci, pos, NULL);
        delete [] pos;
        vector<MapData *> maps; maps.push_back(map);
        vector<GenoLikeData *> *gl = readTGLSData(path, nloci, nind, &maps, gl_type);
        for (int l = 0; l < nloci; l++)
            memcpy(out + (size_t)l * nind, gl->at(0)->data[l], sizeof(double) * nind);
        releaseGLData(gl);
        free_map(map);
    } catch (...) { return 1; }
    return 0;
}

// win: double[nind][nloci] for ONE chromosome; returns count, fills out (capacity nind*nloci)
REF_API int ref_flatten(int nloci, int nind, const double *win, int step, double *out)
{
    try {
        WinData *w = initWinData((unsigned)nind, (unsigned)nloci);
        for (int i = 0; i < nind; i++)
            memcpy(w->data[i], win + (size_t)i * nloci, sizeof(double) * nloci);
        vector<WinData *> v; v.push_back(w);
        DoubleData *d = convertWinData2DoubleData(&v, step);
        int n = d->size;
        memcpy(out, d->data, sizeof(double) * n);
        releaseDoubleData(d);
        releaseWinData(w);
        return n;
    } catch (...) { return -1; }
}
