// Host-side mirror of the reference's Phase-I interface, on top of the C ABI (include/garlic_hip.h).
//
// The reference is a C++ program with no plugin layer: its boundary for this path is
//     calcLODWindows   src/garlic-roh.h:96-102
//     calcwLODWindows  src/garlic-roh.h:104-112
// and the structs those borrow (src/garlic-data.h:32-108).  This header re-declares structs of
// the same shape and the two entry points with the same argument meaning and error behaviour
// (`throw 0`, src/garlic-data.cpp:1619), so GARLIC's callers -- main, exploreWinsizes,
// selectWinsize, selectWinsizeFromList -- can be pointed here unchanged.
// Also: the ingest GARLIC does before the path (tped/tfam/tgls/freq/map/centromere) and the
// two consumers right after it (raw LOD writer, KDE feed flattening).
#pragma once
#include <map>
#include <string>
#include <vector>

namespace garlic_host {

const int MISSING = -9999; // src/garlic-data.h:24

struct HapData {           // src/garlic-data.h:32-38
    short **data;          // [locus][ind]: copies of the counted allele, -9 = missing
    int nind;
    int nloci;
    bool **firstCopy;      // --phased only (else NULL): the first allele of the pair is the counted one
    // Extension: rows straight from the genotype cache, 4 genotypes per byte (0/1/2, 3 = missing),
    // (nind + 3) / 4 bytes each; then `data` is NULL.  Everything in this adapter takes either form
    // (genotypeAt reads one value); the engine uploads the packed rows as they are.
    unsigned char **packed;
};
inline short genotypeAt(const HapData *h, int locus, int ind)
{
    if (h->data) return h->data[locus][ind];
    const unsigned code = (h->packed[locus][ind >> 2] >> (2 * (ind & 3))) & 3u;
    return code == 3u ? (short)-9 : (short)code;
}
struct MapData {           // src/garlic-data.h:51-60
    int *physicalPos;
    double *geneticPos;
    std::string *locusName;
    char *allele;          // the allele that is counted
    int nloci;
    std::string chr;
};
struct IndData {           // src/garlic-data.h:62-67
    std::string pop;
    std::string *indID;
    int nind;
};
struct FreqData {          // src/garlic-data.h:69-73
    double *freq;
    int nloci;
};
struct GenoLikeData {      // src/garlic-data.h:89-95
    double **data;         // [locus][ind]: per-genotype error probability
    int nind;
    int nloci;
    // Extension (readTGLSData with compact = true): the same values dictionary-coded, one byte per
    // genotype -- codes[locus][ind] indexes values[0 .. nvalues) -- and `data` NULL.  GQ / PL / GL
    // files hold a few dozen distinct values; the engine uploads the codes as they are.
    unsigned char **codes;
    double *values;
    int nvalues;
};
inline double likelihoodAt(const GenoLikeData *g, int locus, int ind)
{
    return g->data ? g->data[locus][ind] : g->values[g->codes[locus][ind]];
}
struct LDData {            // src/garlic-data.h:103-108
    double **LD;           // [locus][winsize]
    int nloci;
    int winsize;
};
struct GenoFreqData {      // src/garlic-data.h:75-79
    double *homFreq;       // fraction of homozygous genotypes among the non-missing, per locus
    int nloci;
};
struct WinData {           // src/garlic-data.h:81-87
    double **data;         // [ind][locus], MISSING where no score
    int nind;
    int nloci;
};
struct DoubleData {        // src/garlic-data.h:97-101
    double *data;
    int size;
};

// src/garlic-centromeres.h: only centromereStart/End are used on the path (garlic-roh.cpp:36-37)
class centromere {
public:
    centromere() {}
    // arg: hg18 | hg19 | hg38 | none ; file: custom "chr start end" table (garlic-centromeres.cpp:64)
    centromere(const std::string &arg, const std::string &file, const std::string &defaultFileName);
    int centromereStart(const std::string &chr);
    int centromereEnd(const std::string &chr);
    void readCustomCentromeres(const std::string &filename);
    void set(const std::string &chr, int start, int end);

private:
    std::map<std::string, int> gapStart, gapEnd;
    std::map<std::string, int> warned;
};

std::string checkChrName(std::string chr); // "1" -> "chr1" (garlic-data.cpp:1886-1891)

// ---- allocation helpers with the reference's semantics
HapData *initHapData(unsigned int nind, unsigned int nloci, bool PHASED = false);   // garlic-data.cpp:1749
void releaseHapData(HapData *d);
void releaseHapData(std::vector<HapData *> *v);
MapData *initMapData(int nloci);
void releaseMapData(MapData *d);
void releaseMapData(std::vector<MapData *> *v);
FreqData *initFreqData(int nloci);
void releaseFreqData(FreqData *d);
void releaseFreqData(std::vector<FreqData *> *v);
GenoLikeData *initGLData(unsigned int nind, unsigned int nloci);
void releaseGLData(GenoLikeData *d);
void releaseGLData(std::vector<GenoLikeData *> *v);
LDData *initLDData(int nloci, int winsize);
void releaseLDData(LDData *d);
void releaseLDData(std::vector<LDData *> *v);
std::vector<GenoFreqData *> *calculateGenoFreq(std::vector<HapData *> *hapDataByChr);   // garlic-data.cpp:648
void releaseGenoFreq(std::vector<GenoFreqData *> *v);
WinData *initWinData(unsigned int nind, unsigned int nloci);          // throws 0 on empty shapes
std::vector<WinData *> *initWinData(std::vector<MapData *> *mapDataByChr, int nind);
void releaseWinData(WinData *d);
void releaseWinData(std::vector<WinData *> *v);
void releaseDoubleData(DoubleData *d);
void releaseIndData(IndData *d);

// ---- ingest (what main does before the path, src/garlic-main.cpp:216-279)
void loadTPEDData(const std::string &tpedfile, int &numLoci, int &numInd,
                  std::vector<HapData *> **hapDataByChr, std::vector<MapData *> **mapDataByChr,
                  std::vector<FreqData *> **freqDataByChr, char TPED_MISSING,
                  bool PHASED = false, int nresample = 0,
                  unsigned long long resampleSeed = 0);                          // garlic-data.cpp:10
// nresample > 0 (--resample, garlic-data.cpp:142-148): binomial resampling of every frequency, one mt19937
// stream in file order; resampleSeed 0 = time(NULL) as in the reference, else the stream GSL gives that seed
void scanIndData3(const std::string &filename, int &numInd, std::string &popName);  // :1893
IndData *readIndData3(const std::string &filename, int numInd);                     // :1963
std::vector<GenoLikeData *> *readTGLSData(const std::string &filename, int expectedLoci, int expectedInd,
                                          std::vector<MapData *> *mapDataByChr,
                                          const std::string &GL_TYPE,
                                          bool compact = false);   // compact: ::codes instead of ::data             // :1516
std::vector<FreqData *> *readFreqData(const std::string &freqfile,
                                      std::vector<MapData *> *mapDataByChr);        // :1345
void writeFreqData(const std::string &freqOutfile, std::vector<FreqData *> *freqDataByChr,
                   std::vector<MapData *> *mapDataByChr);                           // :1442
int filterMonomorphicSites(std::vector<MapData *> **mapDataByChr, std::vector<HapData *> **hapDataByChr,
                           std::vector<FreqData *> **freqDataByChr,
                           std::vector<GenoLikeData *> **GLDataByChr, bool USE_GL); // :871
// genetic map for --weighted: scaffold of 4 columns chr snpid gpos ppos (:760-844), out-of-bounds /
// centromere / monomorphic filter (:916-960, 1066-1098), linear interpolation (:702-757)
struct GenMapScaffold {
    std::vector<int> physicalPos;
    std::vector<double> geneticPos;
    std::string chr;
    int centroStart = 0, centroEnd = 0;
};
std::vector<GenMapScaffold *> *loadMapScaffold(const std::string &mapfile, centromere *centro);
void releaseGenMapScaffold(std::vector<GenMapScaffold *> *v);
int filterMonomorphicAndOOBSites(std::vector<MapData *> **mapDataByChr, std::vector<HapData *> **hapDataByChr,
                                 std::vector<FreqData *> **freqDataByChr,
                                 std::vector<GenoLikeData *> **GLDataByChr,
                                 std::vector<GenMapScaffold *> *scaffoldMapByChr, bool USE_GL);
int interpolateGeneticmap(std::vector<MapData *> *mapDataByChr, std::vector<GenMapScaffold *> *scaffoldMapByChr);

// Binary sidecar of what loadTPEDData produces: parsing a 10M x 10k TPED is ~400 GB
// of text and dwarfs the GPU time; the cache holds the same genotypes at 2 bits each (4 per byte,
// SNP-major; 3 = missing), positions, genetic positions, locus names, counted alleles and
// frequencies, and loads at memory speed; if the genotypes were read with PHASED, also
// HapData::firstCopy at 1 bit each.  Same (hap, map, freq) triple as the TPED path, so everything
// downstream is unchanged.  `throw 0` on I/O or format errors.
void writeGenotypeCache(const std::string &path, std::vector<HapData *> *hapDataByChr,
                        std::vector<MapData *> *mapDataByChr, std::vector<FreqData *> *freqDataByChr);
void loadGenotypeCache(const std::string &path, int &numLoci, int &numInd,
                       std::vector<HapData *> **hapDataByChr, std::vector<MapData *> **mapDataByChr,
                       std::vector<FreqData *> **freqDataByChr,
                       bool keepPacked = false);   // keepPacked: HapData::packed instead of ::data

// ---- ROH calls (src/garlic-roh.h:43-57)
struct ROHData {
    std::string indID;
    std::vector<int> chr;
    std::vector<double> start, stop, length;
};
struct ROHLength {
    std::string pop;
    double *length;
    double size;
};
std::vector<ROHData *> *initROHData(IndData *indData);                 // garlic-roh.cpp:387-397
void releaseROHData(std::vector<ROHData *> *rohDataByInd);             // :399-407
ROHLength *initROHLength(int size, std::string pop);                   // :547-554
void releaseROHLength(ROHLength *rohLength);                           // :556-560
// the .roh.bed file (garlic-roh.cpp:574-648): a track line per individual, then chr / start / stop / size class /
// size / colour per segment; bounds: the size-class boundaries (--size-bounds, or GARLIC's GMM stage)
void writeROHData(const std::string &outfile, std::vector<ROHData *> *rohDataByInd, std::vector<MapData *> *mapDataByChr,
                  const std::vector<double> &bounds, const std::string &popName, const std::string &version, bool CM);

// ---- the path (drop-in signatures)
struct LodOptions {
    std::vector<int> devices;   // HIP device ordinals; empty = {0}.  Individuals shard contiguously.
    // --ld-subsample draw: the reference seeds its generator with time(NULL)
    // (garlic-data.cpp:346); 0 does the same here, any other value makes the draw repeatable.
    unsigned long long ld_seed = 0;
};
void setLodOptions(const LodOptions &o);

std::vector<WinData *> *calcLODWindows(std::vector<HapData *> *hapDataByChr,
                                       std::vector<FreqData *> *freqDataByChr,
                                       std::vector<MapData *> *mapDataByChr,
                                       std::vector<GenoLikeData *> *GLDataByChr, centromere *centro,
                                       int winsize, double error, int MAX_GAP, bool USE_GL);

std::vector<WinData *> *calcwLODWindows(std::vector<HapData *> *hapDataByChr,
                                        std::vector<FreqData *> *freqDataByChr,
                                        std::vector<MapData *> *mapDataByChr,
                                        std::vector<GenoLikeData *> *GLDataByChr,
                                        std::vector<LDData *> *ldDataByChr, centromere *centro,
                                        int winsize, double error, int MAX_GAP, bool USE_GL, int M,
                                        double mu, int numThreads);

// LD weights of wLOD (garlic-data.cpp:330-375).  Same arguments as the reference; the counts come
// from the device(s), so genoFreqDataByChr is not read (the device recomputes the same fractions
// from the genotypes) and numThreads is ignored.  PHASED (calcR2LD) needs HapData::firstCopy
// (loadTPEDData with PHASED), else `throw 0`.
std::vector<LDData *> *calcLDData(std::vector<HapData *> *hapDataByChr, std::vector<FreqData *> *freqDataByChr,
                                  std::vector<MapData *> *mapDataByChr,
                                  std::vector<GenoFreqData *> *genoFreqDataByChr, centromere *centro,
                                  int winsize, int MAX_GAP, bool PHASED, int numThreads, int ldSubsample);
// the individuals calcLDData would use: all when ldSubsample <= 0 or >= nind, else ldSubsample
// distinct indices in increasing order (what gsl_ran_choose returns, garlic-data.cpp:361-362)
std::vector<int> drawLdSubsample(int nind, int ldSubsample, unsigned long long seed);
// the individuals convertSubsetWinData2DoubleData would use (garlic-data.cpp:2081-2096; time-seeded
// there): empty = everyone (kdeSubsample <= 0 or >= nind), else kdeSubsample distinct indices, increasing
std::vector<int> drawKdeSubsample(int nind, int kdeSubsample, unsigned long long seed);

// A panel kept on the device(s) across window sizes (exploreWinsizes / selectWinsize call the
// path once per candidate winsize on the same data, garlic-roh.cpp:726-751,798-837,881-920).
class LodEngine {
public:
    LodEngine(std::vector<HapData *> *hapDataByChr, std::vector<FreqData *> *freqDataByChr,
              std::vector<MapData *> *mapDataByChr, std::vector<GenoLikeData *> *GLDataByChr,
              centromere *centro, bool USE_GL, const std::vector<int> &devices);
    ~LodEngine();
    std::vector<WinData *> *lodWindows(int winsize, double error, int MAX_GAP);
    std::vector<WinData *> *wlodWindows(std::vector<LDData *> *ldDataByChr, int winsize, double error,
                                        int MAX_GAP, int M, double mu);
    // LD weights for winsize from the resident genotypes (subsample: panel-wide individual indices,
    // empty = all); they stay installed on the device(s) for wlodWindowsResident.  Returns the
    // reference-shaped host copy when want_host is set, else NULL.
    // phased: calcR2LD (r2 from HapData::firstCopy and FreqData::freq) instead of calcHR2LD.
    std::vector<LDData *> *ldWeights(int winsize, const std::vector<int> &subsample, bool want_host = true,
                                     bool phased = false);
    std::vector<WinData *> *wlodWindowsResident(int winsize, double error, int MAX_GAP, int M, double mu);
    // What exploreWinsizes / selectWinsize keep of a window size (garlic-roh.cpp:741-745, 816-823):
    // convertWinData2DoubleData(calcLODWindows(...), step), with the scores thinned on the device(s)
    // -- 8/step bytes per window come back instead of 8.  weighted: wLOD from the resident LD weights.
    // kdeSubsample (convertSubsetWinData2DoubleData, garlic-data.cpp:2071-2150: selectLODCutoff with
    // --kde-subsample): the feed of these individuals only, panel-wide indices in increasing order (what
    // gsl_ran_choose draws); NULL or empty = everyone.  Only their 64-individual blocks are scored.
    DoubleData *lodFeed(int winsize, double error, int MAX_GAP, int step, bool weighted = false, int M = 0,
                        double mu = 0.0, const std::vector<int> *kdeSubsample = nullptr);
    // several window sizes in one call (unweighted --error scores): the feeds of exploreWinsizes / selectWinsize /
    // selectWinsizeFromList (garlic-roh.cpp:726-751, 798-837, 881-920), one DoubleData per size; steps NULL: the sizes
    std::vector<DoubleData *> lodFeedMulti(const std::vector<int> &winsizes, double error, int MAX_GAP,
                                           const std::vector<int> *steps = nullptr,
                                           const std::vector<int> *kdeSubsample = nullptr);
    // assembleROHWindows (garlic-roh.cpp:409-545) for the resident panel, from the genotypes to rohData->start / stop /
    // length without the window scores or the per-SNP coverage counts in memory (garlic_roh_segments): what main needs of
    // calcLODWindows + assembleROHWindows when --raw-lod is not asked for (garlic-main.cpp:346-420).  Same arguments as
    // the reference's function after the data; weighted: wLOD from the resident LD weights (ldWeights).
    std::vector<ROHData *> *assembleROHWindows(IndData *indData, double lodScoreCutoff, ROHLength **rohLength, int winSize,
                                               double error, int MAX_GAP, double OVERLAP_FRAC, bool CM, bool weighted = false,
                                               int M = 0, double mu = 0.0);
    LodEngine(const LodEngine &) = delete;
    LodEngine &operator=(const LodEngine &) = delete;

private:
    std::vector<WinData *> *run(bool weighted, int winsize, double error, int MAX_GAP, int M, double mu);
    void upload(std::vector<HapData *> *haps, std::vector<FreqData *> *freqs, std::vector<MapData *> *maps,
                std::vector<GenoLikeData *> *gls, centromere *centro, bool USE_GL, const std::vector<int> &devices);
    void release();
    struct Impl;
    Impl *impl;
};

// ---- consumers right after the path
DoubleData *convertWinData2DoubleData(std::vector<WinData *> *winDataByChr, int step); // :2026
// :2071 with the drawn individuals (randInd[], any order) supplied instead of a time-seeded gsl_ran_choose
DoubleData *convertSubsetWinData2DoubleData(std::vector<WinData *> *winDataByChr, const std::vector<int> &randInd,
                                            int step);
void writeWinData(std::vector<WinData *> *winDataByChr, IndData *indData,
                  std::vector<MapData *> *mapDataByChr, const std::string &outfile);      // :1704

} // namespace garlic_host
