// Host adapter: the reference's Phase-I interface on top of libgarlic_hip.so (see garlic_host.hpp).
#include "garlic_host.hpp"

#include "../../include/garlic_hip.h"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <random>
#include <mutex>
#include <thread>
#include <ctime>

namespace garlic_host {

namespace {

struct CentroRow {
    const char *chr;
    int start, end;
};
#include "centromere_tables.inc"

[[noreturn]] void fail(const std::string &msg)
{
    std::cerr << "ERROR: " << msg << "\n";
    throw 0; // the reference's error convention (src/garlic-data.cpp:1619; main catches `...`)
}

// line reader over plain or gzip files (the reference reads everything through gzstream)
class LineReader {
public:
    explicit LineReader(const std::string &path) : f(gzopen(path.c_str(), "rb"))
    {
        if (!f) fail("Failed to open " + path);
        gzbuffer(f, 1 << 20);
    }
    ~LineReader() { if (f) gzclose(f); }
    bool next(std::string &line)
    {
        line.clear();
        char buf[1 << 16];
        while (gzgets(f, buf, sizeof buf)) {
            size_t n = strlen(buf);
            if (n && buf[n - 1] == '\n') {
                line.append(buf, n - 1);
                if (!line.empty() && line.back() == '\r') line.pop_back();
                return true;
            }
            line.append(buf, n);
        }
        return !line.empty();
    }

private:
    gzFile f;
};

int countFields(const std::string &s)
{
    int n = 0;
    bool in = false;
    for (char c : s) {
        bool ws = (c == ' ' || c == '\t' || c == '\r' || c == '\n');
        if (!ws && !in) n++;
        in = !ws;
    }
    return n;
}

LodOptions g_options;

void check(int rc, const char *what)
{
    if (rc != GARLIC_OK) fail(std::string(what) + ": " + garlic_hip_last_error());
}

} // namespace

// ------------------------------------------------------------------------- centromere
std::string checkChrName(std::string chr)
{
    if (chr.empty() || chr[0] != 'c') chr = "chr" + chr;
    return chr;
}

centromere::centromere(const std::string &arg, const std::string &file, const std::string &defaultFileName)
{
    const CentroRow *rows = nullptr;
    size_t n = 0;
    if (arg == "hg18") { rows = k_hg18; n = sizeof k_hg18 / sizeof *k_hg18; }
    else if (arg == "hg19") { rows = k_hg19; n = sizeof k_hg19 / sizeof *k_hg19; }
    else if (arg == "hg38") { rows = k_hg38; n = sizeof k_hg38 / sizeof *k_hg38; }
    for (size_t i = 0; i < n; i++) set(rows[i].chr, rows[i].start, rows[i].end);
    if (!rows && file != defaultFileName) readCustomCentromeres(file);
}

void centromere::set(const std::string &chr, int start, int end)
{
    gapStart[checkChrName(chr)] = start;
    gapEnd[checkChrName(chr)] = end;
}

void centromere::readCustomCentromeres(const std::string &filename)
{
    LineReader in(filename);
    std::string line;
    int rows = 0;
    while (in.next(line)) {
        if (countFields(line) != 3) { std::cerr << "ERROR: Custom centromere file requires three columns.\n"; continue; }
        std::stringstream ss(line);
        std::string chr;
        int s, e;
        ss >> chr >> s >> e;
        set(chr, s, e);
        rows++;
    }
    std::cerr << "Loaded custom centromere limits for " << rows << " chromosomes.\n";
}

int centromere::centromereStart(const std::string &chr)
{
    auto it = gapStart.find(chr);
    if (it == gapStart.end()) { // unknown chromosome: 0 and one warning (garlic-centromeres.cpp:33-45)
        if (!warned[chr]++) std::cerr << "WARNING: No centromere information for chr: " << chr << "\n";
        return 0;
    }
    return it->second;
}

int centromere::centromereEnd(const std::string &chr)
{
    auto it = gapEnd.find(chr);
    if (it == gapEnd.end()) {
        if (!warned[chr]++) std::cerr << "WARNING: No centromere information for chr: " << chr << "\n";
        return 0;
    }
    return it->second;
}

// ------------------------------------------------------------------------- structs
HapData *initHapData(unsigned int nind, unsigned int nloci, bool PHASED)
{
    if (nind < 1 || nloci < 1) fail("Can not allocate HapData object: counts must be positive.");
    HapData *d = new HapData;
    d->nind = nind;
    d->nloci = nloci;
    d->data = new short *[nloci];
    d->firstCopy = PHASED ? new bool *[nloci] : nullptr;
    d->packed = nullptr;
    for (unsigned l = 0; l < nloci; l++) {
        d->data[l] = new short[nind];
        std::fill(d->data[l], d->data[l] + nind, (short)MISSING);
        if (PHASED) {
            d->firstCopy[l] = new bool[nind];
            std::fill(d->firstCopy[l], d->firstCopy[l] + nind, false);
        }
    }
    return d;
}
void releaseHapData(HapData *d)
{
    if (!d) return;
    for (int l = 0; l < d->nloci; l++) {
        if (d->data) delete[] d->data[l];
        if (d->firstCopy) delete[] d->firstCopy[l];
        if (d->packed) delete[] d->packed[l];
    }
    delete[] d->data;
    delete[] d->firstCopy;
    delete[] d->packed;
    delete d;
}
void releaseHapData(std::vector<HapData *> *v) { for (auto d : *v) releaseHapData(d); delete v; }

MapData *initMapData(int nloci)
{
    if (nloci < 1) fail("number of loci must be positive.");
    MapData *d = new MapData;
    d->nloci = nloci;
    d->physicalPos = new int[nloci];
    d->geneticPos = new double[nloci];
    d->locusName = new std::string[nloci];
    d->allele = new char[nloci];
    d->chr = "--";
    for (int l = 0; l < nloci; l++) { d->physicalPos[l] = MISSING; d->geneticPos[l] = MISSING; d->locusName[l] = "--"; d->allele[l] = '-'; }
    return d;
}
void releaseMapData(MapData *d)
{
    if (!d) return;
    delete[] d->physicalPos; delete[] d->geneticPos; delete[] d->locusName; delete[] d->allele;
    delete d;
}
void releaseMapData(std::vector<MapData *> *v) { for (auto d : *v) releaseMapData(d); delete v; }

FreqData *initFreqData(int nloci)
{
    if (nloci < 1) fail("number of loci must be positive.");
    FreqData *d = new FreqData;
    d->nloci = nloci;
    d->freq = new double[nloci];
    std::fill(d->freq, d->freq + nloci, (double)MISSING);
    return d;
}
void releaseFreqData(FreqData *d) { if (d) { delete[] d->freq; delete d; } }
void releaseFreqData(std::vector<FreqData *> *v) { for (auto d : *v) releaseFreqData(d); delete v; }

GenoLikeData *initGLData(unsigned int nind, unsigned int nloci)
{
    if (nind < 1 || nloci < 1) fail("Can not allocate GenoLikeData object: counts must be positive.");
    GenoLikeData *d = new GenoLikeData;
    d->nind = nind;
    d->nloci = nloci;
    d->data = new double *[nloci];
    d->codes = nullptr;
    d->values = nullptr;
    d->nvalues = 0;
    for (unsigned l = 0; l < nloci; l++) {
        d->data[l] = new double[nind];
        std::fill(d->data[l], d->data[l] + nind, (double)MISSING);
    }
    return d;
}
void releaseGLData(GenoLikeData *d)
{
    if (!d) return;
    for (int l = 0; l < d->nloci; l++) {
        if (d->data) delete[] d->data[l];
        if (d->codes) delete[] d->codes[l];
    }
    delete[] d->data;
    delete[] d->codes;
    delete[] d->values;
    delete d;
}
void releaseGLData(std::vector<GenoLikeData *> *v) { for (auto d : *v) releaseGLData(d); delete v; }

LDData *initLDData(int nloci, int winsize)
{
    LDData *d = new LDData;
    d->nloci = nloci;
    d->winsize = winsize;
    d->LD = new double *[nloci];
    for (int l = 0; l < nloci; l++) {
        d->LD[l] = new double[winsize];
        std::fill(d->LD[l], d->LD[l] + winsize, 0.0);
    }
    return d;
}
void releaseLDData(LDData *d)
{
    if (!d) return;
    for (int l = 0; l < d->nloci; l++) delete[] d->LD[l];
    delete[] d->LD;
    delete d;
}

void releaseLDData(std::vector<LDData *> *v)
{
    if (!v) return;
    for (auto d : *v) releaseLDData(d);
    delete v;
}

std::vector<GenoFreqData *> *calculateGenoFreq(std::vector<HapData *> *haps)
{   // garlic-data.cpp:648-676
    std::vector<GenoFreqData *> *out = new std::vector<GenoFreqData *>;
    for (auto h : *haps) {
        GenoFreqData *g = new GenoFreqData;
        g->nloci = h->nloci;
        g->homFreq = new double[h->nloci];
        for (int l = 0; l < h->nloci; l++) {
            double total = 0, hom = 0;
            for (int i = 0; i < h->nind; i++) {
                const short v = genotypeAt(h, l, i);
                if (v != -9) {
                    if (v == 2 || v == 0) hom++;
                    total++;
                }
            }
            hom /= total;
            g->homFreq[l] = hom;
        }
        out->push_back(g);
    }
    return out;
}
void releaseGenoFreq(std::vector<GenoFreqData *> *v)
{
    if (!v) return;
    for (auto g : *v) { delete[] g->homFreq; delete g; }
    delete v;
}

WinData *initWinData(unsigned int nind, unsigned int nloci)
{
    if (nind < 1 || nloci < 1) {
        std::cerr << "ERROR: Can't allocate WinData object.  Number of individuals (" << nind
                  << ") and number of loci (" << nloci << ") must be positive.\n";
        throw 0;
    }
    WinData *d = new WinData;
    d->nind = nind;
    d->nloci = nloci;
    d->data = new double *[nind];
    for (unsigned i = 0; i < nind; i++) d->data[i] = new double[nloci]; // filled by the device result
    return d;
}
std::vector<WinData *> *initWinData(std::vector<MapData *> *maps, int nind)
{
    auto *v = new std::vector<WinData *>;
    for (auto m : *maps) v->push_back(initWinData(nind, m->nloci));
    return v;
}
void releaseWinData(WinData *d)
{
    if (!d) return;
    for (int i = 0; i < d->nind; i++) delete[] d->data[i];
    delete[] d->data;
    delete d;
}
void releaseWinData(std::vector<WinData *> *v) { for (auto d : *v) releaseWinData(d); delete v; }
void releaseDoubleData(DoubleData *d) { if (d) { delete[] d->data; delete d; } }
void releaseIndData(IndData *d) { if (d) { delete[] d->indID; delete d; } }

// ------------------------------------------------------------------------- ingest
namespace {
void flushChromosome(const std::string &chr, std::vector<short *> &hap, std::vector<bool *> &fc, std::vector<double> &gpos,
                     std::vector<double> &ppos, std::vector<std::string> &names, std::vector<char> &allele,
                     std::vector<double> &freq, int numInd, std::vector<HapData *> *haps,
                     std::vector<MapData *> *maps, std::vector<FreqData *> *freqs)
{
    const int n = (int)hap.size();
    MapData *m = initMapData(n);
    m->chr = checkChrName(chr);
    HapData *h = new HapData;
    h->nind = numInd;
    h->nloci = n;
    h->data = new short *[n];
    h->firstCopy = fc.empty() ? nullptr : new bool *[n];
    h->packed = nullptr;
    FreqData *f = initFreqData(n);
    for (int l = 0; l < n; l++) {
        if (h->firstCopy) h->firstCopy[l] = fc[l];
        m->physicalPos[l] = (int)ppos[l]; // read as double, stored as int (garlic-data.cpp:41,229)
        m->geneticPos[l] = gpos[l];
        m->locusName[l] = names[l];
        m->allele[l] = allele[l];
        h->data[l] = hap[l];
        f->freq[l] = freq[l];
    }
    maps->push_back(m);
    haps->push_back(h);
    freqs->push_back(f);
    hap.clear(); fc.clear(); gpos.clear(); ppos.clear(); names.clear(); allele.clear(); freq.clear();
}
} // namespace

namespace {
struct TpedLine {          // one parsed TPED line (garlic-data.cpp:56-141)
    std::string chr, name;
    double g = 0, p = 0;
    int numInd = 0;
    short *data = nullptr;
    bool *first = nullptr;
    char one = 0;
    double freq = 0;
    int total = 0;         // non-missing alleles on the line
};

TpedLine parseTpedLine(const std::string &line, char TPED_MISSING, bool PHASED)
{
    TpedLine r;
    r.numInd = (countFields(line) - 4) / 2; // garlic-data.cpp:59-60
    // the four leading fields through a stream (tiny), the allele columns by hand: at 10k
    // individuals a line is 40 KB and formatted extraction of every character dominated the load
    size_t at = 0;
    for (int field = 0; field < 4; field++) {
        while (at < line.size() && isspace((unsigned char)line[at])) at++;
        while (at < line.size() && !isspace((unsigned char)line[at])) at++;
    }
    std::stringstream ss(line.substr(0, at));
    ss >> r.chr >> r.name >> r.g >> r.p;
    const char *cur = line.data() + at, *const end = line.data() + line.size();
    auto next_allele = [&]() -> char {   // `ss >> char`: the next non-blank character, TPED_MISSING at the end
        while (cur < end && (*cur == ' ' || *cur == '\t' || *cur == '\r')) cur++;
        return cur < end ? *cur++ : TPED_MISSING;
    };
    const int n = std::max(r.numInd, 0);
    r.data = new short[n > 0 ? n : 1];
    r.first = PHASED ? new bool[n > 0 ? n : 1] : nullptr;
    char one = TPED_MISSING; // the first non-missing allele on the line is the counted one
    int nalleles = 0, total = 0;
    for (int i = 0; i < n; i++) {
        const char a1 = next_allele(), a2 = next_allele(); // alleles are single characters (garlic-data.cpp:47,111)
        if (one == TPED_MISSING && a1 != TPED_MISSING) one = a1;
        if (one == TPED_MISSING && a2 != TPED_MISSING) one = a2;
        int v = 0;
        for (char a : {a1, a2}) {
            if (a == TPED_MISSING) v += -9;
            else { total++; if (a == one) { v += 1; nalleles++; } }
        }
        r.data[i] = (short)(v < 0 ? -9 : v);
        if (r.first) r.first[i] = (a1 == one);   // garlic-data.cpp:129
    }
    r.one = one;
    r.total = total;
    r.freq = total == 0 ? 0.0 : double(nalleles) / double(total); // garlic-data.cpp:140-141
    return r;
}
} // namespace

void loadTPEDData(const std::string &tpedfile, int &numLoci, int &numInd, std::vector<HapData *> **hapDataByChr,
                  std::vector<MapData *> **mapDataByChr, std::vector<FreqData *> **freqDataByChr,
                  char TPED_MISSING, bool PHASED, int nresample, unsigned long long resampleSeed)
{
    // --resample (garlic-data.cpp:16-20, 142-148): the frequency of every SNP becomes the fraction of
    // nresample uniform draws that fall at or below it, one generator for the whole file, in file order.
    // The reference's generator is GSL's default, mt19937 seeded with time(NULL), gsl_rng_uniform =
    // 32 bits / 2^32: std::mt19937 seeded alike is the same stream (GSL maps seed 0 to 4357), so a run
    // given the reference's seed resamples to the same frequencies.
    const unsigned long long seed0 = resampleSeed ? resampleSeed : (unsigned long long)time(nullptr);
    std::mt19937 resampler((uint32_t)(seed0 & 0xffffffffull) ? (uint32_t)(seed0 & 0xffffffffull) : 4357u);
    LineReader in(tpedfile);
    *hapDataByChr = new std::vector<HapData *>;
    *mapDataByChr = new std::vector<MapData *>;
    *freqDataByChr = new std::vector<FreqData *>;
    std::vector<short *> hap;
    std::vector<bool *> fc;
    std::vector<double> gpos, ppos, freq;
    std::vector<std::string> names;
    std::vector<char> allele;
    std::string chr, prevChr;
    numLoci = 0;
    numInd = 0;
    // Lines are independent: one thread reads (and inflates), batches of lines are parsed on all
    // cores, the results join the chromosomes in file order.
    const unsigned nthreads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    const size_t batch_lines = 64 * nthreads;
    std::vector<std::string> lines;
    std::vector<TpedLine> parsed;
    bool more = true;
    while (more) {
        lines.clear();
        std::string line;
        while (lines.size() < batch_lines && (more = in.next(line)))
            if (!line.empty()) lines.push_back(std::move(line));
        parsed.assign(lines.size(), TpedLine());
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nthreads && t < lines.size(); t++)
            th.emplace_back([&, t] {
                for (size_t k = t; k < lines.size(); k += nthreads) parsed[k] = parseTpedLine(lines[k], TPED_MISSING, PHASED);
            });
        for (auto &t : th) t.join();
        for (TpedLine &r : parsed) {
            numLoci++;
            numInd = r.numInd;
            chr = r.chr;
            if (numLoci == 1) prevChr = chr;
            if (chr != prevChr) { // new chromosome when the chr string changes (garlic-data.cpp:68-91)
                flushChromosome(prevChr, hap, fc, gpos, ppos, names, allele, freq, numInd, *hapDataByChr,
                                *mapDataByChr, *freqDataByChr);
                prevChr = chr;
            }
            hap.push_back(r.data);
            if (r.first) fc.push_back(r.first);
            gpos.push_back(r.g);
            ppos.push_back(r.p);
            names.push_back(r.name);
            allele.push_back(r.one);
            if (nresample > 0 && r.total != 0) {
                int count = 0;
                for (int i = 0; i < nresample; i++)
                    if ((double)resampler() / 4294967296.0 <= r.freq) count++;
                r.freq = double(count) / double(nresample);
            }
            freq.push_back(r.freq);
        }
    }
    if (numLoci == 0) fail("no loci in " + tpedfile);
    flushChromosome(chr, hap, fc, gpos, ppos, names, allele, freq, numInd, *hapDataByChr, *mapDataByChr,
                    *freqDataByChr);
}

void scanIndData3(const std::string &filename, int &numInd, std::string &popName)
{
    LineReader in(filename);
    std::map<std::string, int> seen;
    std::string line, pop, ind;
    int n = 0;
    while (in.next(line)) {
        n++;
        if (countFields(line) < 2) fail("line " + std::to_string(n) + " of " + filename + " has fewer than 2 columns");
        std::stringstream ss(line);
        ss >> pop >> ind;
        if (seen.count(ind)) fail("Found duplicate individual ID (" + ind + ") in " + filename);
        seen[ind] = 1;
        if (n == 1) popName = pop;
        else if (pop != popName) fail("Found multiple population IDs (" + pop + ", " + popName + ") in " + filename);
    }
    numInd = n;
}

IndData *readIndData3(const std::string &filename, int numInd)
{
    if (numInd < 1) fail("Number of individuals must be positive");
    LineReader in(filename);
    IndData *d = new IndData;
    d->nind = numInd;
    d->indID = new std::string[numInd];
    std::string line, pop, ind;
    for (int i = 0; i < numInd && in.next(line); i++) {
        std::stringstream ss(line);
        ss >> pop >> ind;
        d->indID[i] = ind;
    }
    d->pop = pop;
    return d;
}

std::vector<GenoLikeData *> *readTGLSData(const std::string &filename, int /*expectedLoci*/, int expectedInd,
                                          std::vector<MapData *> *maps, const std::string &GL_TYPE, bool compact)
{
    if (GL_TYPE != "GQ" && GL_TYPE != "GL" && GL_TYPE != "PL") fail("Must choose GQ/GL/PL for genotype likelihood format");
    LineReader in(filename);
    auto *out = new std::vector<GenoLikeData *>;
    std::string line, junk;
    for (auto m : *maps) {
        GenoLikeData *d;
        if (compact) {   // one byte per genotype and a table of the distinct converted values
            d = new GenoLikeData{nullptr, expectedInd, m->nloci, new unsigned char *[m->nloci](), new double[256], 0};
        } else {
            d = initGLData(expectedInd, m->nloci);
        }
        out->push_back(d);
        for (int l = 0; l < m->nloci; l++) {
            if (!in.next(line)) fail("too few lines in " + filename);
            const int num = countFields(line);
            if (num != expectedInd + 4) fail("Incorrect number of columns in tgls file: " + std::to_string(num));
            std::stringstream ss(line);
            ss >> junk >> junk >> junk >> junk;
            for (int i = 0; i < expectedInd; i++) {
                double gl;
                ss >> gl;
                // garlic-data.cpp:1557-1576, operation for operation
                if (GL_TYPE == "GQ") { gl /= (-10.0); gl = (gl > -10) ? gl : -10; gl = pow(10, gl); }
                else if (GL_TYPE == "GL") { gl = (gl > -10) ? gl : -10; gl = 1 - pow(10, gl); }
                else { gl /= (-10.0); gl = (gl > -10) ? gl : -10; gl = 1 - pow(10, gl); }
                if (gl <= 0) gl = 0.0000000000000001;
                if (gl > 1) gl = 1;
                if (!compact) { d->data[l][i] = gl; continue; }
                if (i == 0) d->codes[l] = new unsigned char[expectedInd];
                int code = 0;
                while (code < d->nvalues && memcmp(&d->values[code], &gl, sizeof gl) != 0) code++;
                if (code == d->nvalues) {
                    if (code == 256) fail("more than 256 distinct genotype likelihood values on " + m->chr + " in " + filename);
                    d->values[d->nvalues++] = gl;
                }
                d->codes[l][i] = (unsigned char)code;
            }
        }
    }
    return out;
}

std::vector<FreqData *> *readFreqData(const std::string &freqfile, std::vector<MapData *> *maps)
{
    LineReader in(freqfile);
    auto *out = new std::vector<FreqData *>;
    std::string line, chrom, id;
    in.next(line); // header
    int lineNum = 1;
    for (auto m : *maps) {
        FreqData *f = initFreqData(m->nloci);
        out->push_back(f);
        for (int l = 0; l < m->nloci; l++) {
            lineNum++;
            if (!in.next(line)) fail("at line " + std::to_string(lineNum) + " in " + freqfile + ". Perhaps too few lines?");
            if (countFields(line) < 5) fail("fewer than 5 columns in " + freqfile + " on line " + std::to_string(lineNum));
            std::stringstream ss(line);
            double pos;
            char al;
            ss >> chrom >> id >> pos >> al >> f->freq[l];
            if (m->locusName[l] != id) fail("Loci appear mismatched in: " + freqfile + " at line " + std::to_string(lineNum));
            if (m->allele[l] != al) f->freq[l] = 1 - f->freq[l]; // other allele counted (garlic-data.cpp:1422-1424)
        }
    }
    return out;
}

static std::string gzip_member(const std::string &text);

// The reference's table (garlic-data.cpp:790-826; default ostream formatting = 6 significant digits = %g), lines
// formatted and compressed on all host cores, 100k lines per gzip member, appended in order.
void writeFreqData(const std::string &freqOutfile, std::vector<FreqData *> *freqs, std::vector<MapData *> *maps)
{
    const std::string path = freqOutfile + ".gz";
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) fail("Failed to open " + path);
    struct Chunk { size_t c; int l0, l1; };
    std::vector<Chunk> chunks;
    for (size_t c = 0; c < maps->size(); c++)
        for (int l0 = 0; l0 < maps->at(c)->nloci; l0 += 100000)
            chunks.push_back(Chunk{c, l0, std::min(maps->at(c)->nloci, l0 + 100000)});
    const unsigned nthreads = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    std::atomic<bool> failed{false};   // set by the compression threads
    {
        const std::string m = gzip_member("CHR\tSNP\tPOS\tALLELE\tFREQ\n");
        failed = fwrite(m.data(), 1, m.size(), f) != m.size();
    }
    for (size_t k0 = 0; k0 < chunks.size() && !failed; k0 += nthreads) {
        const size_t n = std::min<size_t>(nthreads, chunks.size() - k0);
        std::vector<std::string> members(n);
        std::vector<std::thread> pool;
        for (size_t t = 0; t < n; t++)
            pool.emplace_back([&, t]() {
                const Chunk ch = chunks[k0 + t];
                const MapData *m = maps->at(ch.c);
                const double *fr = freqs->at(ch.c)->freq;
                std::string text;
                text.reserve((size_t)(ch.l1 - ch.l0) * 48);
                char num[64];
                for (int l = ch.l0; l < ch.l1; l++) {
                    text += m->chr; text += '\t';
                    text += m->locusName[l]; text += '\t';
                    text.append(num, (size_t)snprintf(num, sizeof num, "%d\t%c\t%g\n", m->physicalPos[l], m->allele[l], fr[l]));
                }
                try { members[t] = gzip_member(text); } catch (...) { failed = true; }
            });
        for (auto &th : pool) th.join();
        for (const std::string &m : members)
            if (!failed && fwrite(m.data(), 1, m.size(), f) != m.size()) failed = true;
    }
    fclose(f);
    if (failed) fail("Failed to write " + path);
}

namespace {
// keeps the SNPs with keep[l] != 0 in map / hap / freq / GL together
void filterSites(size_t c, const std::vector<char> &keep, std::vector<MapData *> *maps, std::vector<HapData *> *haps,
                 std::vector<FreqData *> *freqs, std::vector<GenoLikeData *> *gls)
{
    int n = 0;
    for (char k : keep) n += k != 0;
    MapData *m = maps->at(c);
    HapData *h = haps->at(c);
    FreqData *f = freqs->at(c);
    GenoLikeData *g = gls ? gls->at(c) : nullptr;
    if (n == m->nloci) return;
    if (n < 1) fail("no sites left on " + m->chr + " after filtering");
    MapData *m2 = initMapData(n);
    m2->chr = m->chr;
    HapData *h2 = new HapData{h->data ? new short *[n] : nullptr, h->nind, n, h->firstCopy ? new bool *[n] : nullptr,
                              h->packed ? new unsigned char *[n] : nullptr};
    FreqData *f2 = initFreqData(n);
    GenoLikeData *g2 = nullptr;
    if (g) {
        g2 = new GenoLikeData{g->data ? new double *[n] : nullptr, g->nind, n, g->codes ? new unsigned char *[n] : nullptr,
                              g->values, g->nvalues};
        if (g->values) g->values = nullptr;   // the table moves on with the kept rows
    }
    int j = 0;
    for (int l = 0; l < m->nloci; l++) {
        if (!keep[l]) {
            if (h->data) delete[] h->data[l];
            if (h->packed) delete[] h->packed[l];
            if (h->firstCopy) delete[] h->firstCopy[l];
            if (g && g->data) delete[] g->data[l];
            if (g && g->codes) delete[] g->codes[l];
            continue;
        }
        m2->physicalPos[j] = m->physicalPos[l]; m2->geneticPos[j] = m->geneticPos[l];
        m2->locusName[j] = m->locusName[l]; m2->allele[j] = m->allele[l];
        if (h->data) h2->data[j] = h->data[l];
        if (h->packed) h2->packed[j] = h->packed[l];
        if (h->firstCopy) h2->firstCopy[j] = h->firstCopy[l];
        f2->freq[j] = f->freq[l];
        if (g && g->data) g2->data[j] = g->data[l];
        if (g && g->codes) g2->codes[j] = g->codes[l];
        j++;
    }
    delete[] h->data; delete[] h->firstCopy; delete[] h->packed; delete h;
    if (g) { delete[] g->data; delete[] g->codes; delete g; (*gls)[c] = g2; }
    releaseMapData(m); releaseFreqData(f);
    (*maps)[c] = m2; (*haps)[c] = h2; (*freqs)[c] = f2;
}
} // namespace

int filterMonomorphicSites(std::vector<MapData *> **maps, std::vector<HapData *> **haps,
                           std::vector<FreqData *> **freqs, std::vector<GenoLikeData *> **gls, bool USE_GL)
{
    int total = 0;
    for (size_t c = 0; c < (*maps)->size(); c++) {
        const FreqData *f = (*freqs)->at(c);
        std::vector<char> keep(f->nloci);
        for (int l = 0; l < f->nloci; l++) keep[l] = (f->freq[l] > 0 && f->freq[l] < 1); // garlic-data.cpp:871-914
        filterSites(c, keep, *maps, *haps, *freqs, USE_GL ? *gls : nullptr);
        total += (*maps)->at(c)->nloci;
    }
    return total;
}

// genetic-map scaffold: 4 columns chr snpid gpos ppos, one block per chromosome in file order
// (garlic-data.cpp:760-844)
std::vector<GenMapScaffold *> *loadMapScaffold(const std::string &mapfile, centromere *centro)
{
    std::vector<GenMapScaffold *> *out = new std::vector<GenMapScaffold *>;
    LineReader in(mapfile);
    std::string line, chr, id;
    int n = 0;
    while (in.next(line)) {
        n++;
        if (countFields(line) != 4)
            fail("line " + std::to_string(n) + " of " + mapfile + " does not have the 4 expected columns");
        std::stringstream ss(line);
        double g;
        int p;
        ss >> chr >> id >> g >> p;
        chr = checkChrName(chr);
        if (out->empty() || out->back()->chr != chr) {
            GenMapScaffold *sc = new GenMapScaffold;
            sc->chr = chr;
            sc->centroStart = centro->centromereStart(chr);
            sc->centroEnd = centro->centromereEnd(chr);
            out->push_back(sc);
        }
        out->back()->physicalPos.push_back(p);
        out->back()->geneticPos.push_back(g);
    }
    return out;
}
void releaseGenMapScaffold(std::vector<GenMapScaffold *> *v)
{
    if (!v) return;
    for (auto s : *v) delete s;
    delete v;
}

// --weighted keeps a site iff its frequency is in (0,1), it lies inside the scaffold's span and
// not strictly inside the centromere (garlic-data.cpp:1066-1098)
int filterMonomorphicAndOOBSites(std::vector<MapData *> **maps, std::vector<HapData *> **haps,
                                 std::vector<FreqData *> **freqs, std::vector<GenoLikeData *> **gls,
                                 std::vector<GenMapScaffold *> *scaffold, bool USE_GL)
{
    int total = 0;
    for (size_t c = 0; c < (*maps)->size(); c++) {
        const FreqData *f = (*freqs)->at(c);
        const MapData *m = (*maps)->at(c);
        const GenMapScaffold *sc = scaffold->at(c);
        std::vector<char> keep(f->nloci);
        for (int l = 0; l < f->nloci; l++) {
            const int q = m->physicalPos[l];
            keep[l] = (f->freq[l] > 0 && f->freq[l] < 1) && !(q < sc->physicalPos.front()) &&
                      !(q > sc->physicalPos.back()) && !(q > sc->centroStart && q < sc->centroEnd);
        }
        filterSites(c, keep, *maps, *haps, *freqs, USE_GL ? *gls : nullptr);
        total += (*maps)->at(c)->nloci;
    }
    return total;
}

// exact scaffold positions take the scaffold's value, the others are linearly interpolated between
// their neighbours with the reference's expression (garlic-data.cpp:718-757)
int interpolateGeneticmap(std::vector<MapData *> *maps, std::vector<GenMapScaffold *> *scaffold)
{
    int interpolated = 0;
    for (size_t c = 0; c < maps->size(); c++) {
        MapData *m = maps->at(c);
        const GenMapScaffold *sc = scaffold->at(c);
        const std::vector<int> &pp = sc->physicalPos;
        size_t k = 0;
        for (int l = 0; l < m->nloci; l++) {
            const int q = m->physicalPos[l];
            if (q < pp.front() || q > pp.back())
                fail("Sites outside of map scaffold should have been filtered out.");
            while (k + 1 < pp.size() && pp[k + 1] <= q) k++;
            if (pp[k] == q) { m->geneticPos[l] = sc->geneticPos[k]; continue; }
            const double x0 = pp[k], y0 = sc->geneticPos[k], x1 = pp[k + 1], y1 = sc->geneticPos[k + 1];
            m->geneticPos[l] = (((y1 - y0) / (x1 - x0)) * q + (y0 - ((y1 - y0) / (x1 - x0)) * x0));
            interpolated++;
        }
    }
    return interpolated;
}

// ------------------------------------------------------------------------- genotype cache
namespace {
const char CACHE_MAGIC[8] = {'G', 'A', 'R', 'L', 'I', 'C', '2', 'B'};
const uint32_t CACHE_VERSION = 2;       // 2: a flags word after the chromosome count
const uint32_t CACHE_FLAG_PHASE = 1;    // HapData::firstCopy rows (1 bit each) follow every chromosome's genotypes

struct CacheOut {
    FILE *f;
    explicit CacheOut(const std::string &p) : f(fopen(p.c_str(), "wb")) { if (!f) fail("cannot write " + p); }
    ~CacheOut() { if (f) fclose(f); }
    void put(const void *p, size_t n) { if (n && fwrite(p, 1, n, f) != n) fail("short write to genotype cache"); }
    template <class T> void val(T v) { put(&v, sizeof v); }
    void str(const std::string &s) { val<uint32_t>((uint32_t)s.size()); put(s.data(), s.size()); }
};
struct CacheIn {
    FILE *f;
    explicit CacheIn(const std::string &p) : f(fopen(p.c_str(), "rb")) { if (!f) fail("cannot read " + p); }
    ~CacheIn() { if (f) fclose(f); }
    void get(void *p, size_t n) { if (n && fread(p, 1, n, f) != n) fail("genotype cache is truncated"); }
    template <class T> T val() { T v; get(&v, sizeof v); return v; }
    std::string str()
    {
        const uint32_t n = val<uint32_t>();
        if (n > (1u << 20)) fail("genotype cache is corrupt (string length)");
        std::string s(n, '\0');
        get(&s[0], n);
        return s;
    }
};
} // namespace

void writeGenotypeCache(const std::string &path, std::vector<HapData *> *haps, std::vector<MapData *> *maps,
                        std::vector<FreqData *> *freqs)
{
    CacheOut o(path);
    const int nind = haps->at(0)->nind;
    o.put(CACHE_MAGIC, 8);
    o.val<uint32_t>(CACHE_VERSION);
    o.val<uint32_t>((uint32_t)nind);
    o.val<uint32_t>((uint32_t)maps->size());
    bool phased = true;
    for (auto h : *haps) phased = phased && h->firstCopy;
    o.val<uint32_t>(phased ? CACHE_FLAG_PHASE : 0u);
    const size_t row = ((size_t)nind + 3) / 4, prow = ((size_t)nind + 7) / 8;
    std::vector<uint8_t> bits(row), pbits(prow);
    for (size_t c = 0; c < maps->size(); c++) {
        const MapData *m = maps->at(c);
        const HapData *h = haps->at(c);
        o.str(m->chr);
        o.val<uint32_t>((uint32_t)m->nloci);
        o.put(m->physicalPos, sizeof(int) * m->nloci);
        o.put(m->geneticPos, sizeof(double) * m->nloci);
        o.put(m->allele, (size_t)m->nloci);
        o.put(freqs->at(c)->freq, sizeof(double) * m->nloci);
        for (int l = 0; l < m->nloci; l++) o.str(m->locusName[l]);
        for (int l = 0; l < m->nloci; l++) {
            if (h->packed) { o.put(h->packed[l], row); continue; }
            std::fill(bits.begin(), bits.end(), 0);
            for (int i = 0; i < nind; i++) {
                const short g = h->data[l][i];
                const unsigned code = (g == 0 || g == 1 || g == 2) ? (unsigned)g : 3u;   // -9 = missing
                bits[i >> 2] |= (uint8_t)(code << (2 * (i & 3)));
            }
            o.put(bits.data(), row);
        }
        if (phased)
            for (int l = 0; l < m->nloci; l++) {   // HapData::firstCopy, one bit per individual
                std::fill(pbits.begin(), pbits.end(), 0);
                for (int i = 0; i < nind; i++)
                    if (h->firstCopy[l][i]) pbits[i >> 3] |= (uint8_t)(1u << (i & 7));
                o.put(pbits.data(), prow);
            }
    }
}

void loadGenotypeCache(const std::string &path, int &numLoci, int &numInd, std::vector<HapData *> **haps,
                       std::vector<MapData *> **maps, std::vector<FreqData *> **freqs, bool keepPacked)
{
    CacheIn in(path);
    char magic[8];
    in.get(magic, 8);
    if (memcmp(magic, CACHE_MAGIC, 8) != 0) fail(path + " is not a GARLIC genotype cache");
    if (in.val<uint32_t>() != CACHE_VERSION) fail(path + ": unsupported genotype cache version");
    const int nind = (int)in.val<uint32_t>();
    const uint32_t nchr = in.val<uint32_t>();
    const uint32_t flags = in.val<uint32_t>();
    if (nind < 1 || nchr < 1 || nchr > 100000 || (flags & ~CACHE_FLAG_PHASE))
        fail(path + ": corrupt genotype cache header");
    const bool phased = (flags & CACHE_FLAG_PHASE) != 0;
    *haps = new std::vector<HapData *>;
    *maps = new std::vector<MapData *>;
    *freqs = new std::vector<FreqData *>;
    const size_t row = ((size_t)nind + 3) / 4, prow = ((size_t)nind + 7) / 8;
    std::vector<uint8_t> bits(row), pbits(prow);
    static const short DECODE[4] = {0, 1, 2, -9};
    numLoci = 0;
    try {
        for (uint32_t c = 0; c < nchr; c++) {
            const std::string chr = in.str();
            const int n = (int)in.val<uint32_t>();
            if (n < 1) fail(path + ": corrupt genotype cache (empty chromosome)");
            // each object joins its container as soon as it exists (rows still NULL), so that a
            // file that ends early leaves nothing behind
            MapData *m = initMapData(n);
            (*maps)->push_back(m);
            m->chr = chr;
            in.get(m->physicalPos, sizeof(int) * n);
            in.get(m->geneticPos, sizeof(double) * n);
            in.get(m->allele, (size_t)n);
            FreqData *f = initFreqData(n);
            (*freqs)->push_back(f);
            in.get(f->freq, sizeof(double) * n);
            for (int l = 0; l < n; l++) m->locusName[l] = in.str();
            HapData *h = new HapData{keepPacked ? nullptr : new short *[n](), nind, n, phased ? new bool *[n]() : nullptr,
                                     keepPacked ? new unsigned char *[n]() : nullptr};
            (*haps)->push_back(h);
            for (int l = 0; l < n; l++) {
                if (keepPacked) {   // the rows as stored: the engine uploads them 2-bit
                    h->packed[l] = new unsigned char[row];
                    in.get(h->packed[l], row);
                    continue;
                }
                in.get(bits.data(), row);
                short *d = h->data[l] = new short[nind];
                for (int i = 0; i < nind; i++) d[i] = DECODE[(bits[i >> 2] >> (2 * (i & 3))) & 3];
            }
            for (int l = 0; phased && l < n; l++) {
                in.get(pbits.data(), prow);
                bool *d = h->firstCopy[l] = new bool[nind];
                for (int i = 0; i < nind; i++) d[i] = (pbits[i >> 3] >> (i & 7)) & 1;
            }
            numLoci += n;
        }
    } catch (...) {
        releaseHapData(*haps); releaseMapData(*maps); releaseFreqData(*freqs);
        *haps = nullptr; *maps = nullptr; *freqs = nullptr;
        throw;
    }
    numInd = nind;
}

// ------------------------------------------------------------------------- the path
void setLodOptions(const LodOptions &o) { g_options = o; }

struct LodEngine::Impl {
    struct Shard {
        int device, ind_begin, nind;
        garlic_ctx *ctx = nullptr;
        garlic_panel *panel = nullptr;
    };
    std::vector<Shard> shards;
    std::vector<int32_t> chr_nloci;
    std::vector<MapData *> *maps;
    int nind = 0;
    bool use_gl = false;
    bool have_phase = false;   // every HapData carries firstCopy: the panels hold the phase planes
};

LodEngine::LodEngine(std::vector<HapData *> *haps, std::vector<FreqData *> *freqs, std::vector<MapData *> *maps,
                     std::vector<GenoLikeData *> *gls, centromere *centro, bool USE_GL,
                     const std::vector<int> &devices)
    : impl(new Impl)
{
    try {
        upload(haps, freqs, maps, gls, centro, USE_GL, devices);
    } catch (...) {   // a constructor that throws runs no destructor: release what was created
        release();
        throw;
    }
}

void LodEngine::release()
{
    for (auto &s : impl->shards) {
        if (s.panel) garlic_panel_destroy(s.panel);
        if (s.ctx) garlic_ctx_destroy(s.ctx);
    }
    delete impl;
    impl = nullptr;
}

void LodEngine::upload(std::vector<HapData *> *haps, std::vector<FreqData *> *freqs, std::vector<MapData *> *maps,
                       std::vector<GenoLikeData *> *gls, centromere *centro, bool USE_GL,
                       const std::vector<int> &devices)
{
    const int nchr = (int)maps->size();
    impl->maps = maps;
    impl->nind = haps->at(0)->nind;
    impl->use_gl = USE_GL;
    int64_t nloci = 0;
    for (auto m : *maps) { impl->chr_nloci.push_back(m->nloci); nloci += m->nloci; }
    std::vector<int32_t> pos(nloci), cs(nchr), ce(nchr);
    std::vector<double> gpos(nloci), freq(nloci);
    int64_t o = 0;
    for (int c = 0; c < nchr; c++) {
        const MapData *m = maps->at(c);
        cs[c] = centro->centromereStart(m->chr);   // garlic-roh.cpp:36-37
        ce[c] = centro->centromereEnd(m->chr);
        for (int l = 0; l < m->nloci; l++, o++) {
            pos[o] = m->physicalPos[l];
            gpos[o] = m->geneticPos[l];
            freq[o] = freqs->at(c)->freq[l];
        }
    }
    std::vector<int> devs = devices.empty() ? std::vector<int>{0} : devices;
    const int nd = std::min<int>((int)devs.size(), impl->nind);
    const int per = (impl->nind + nd - 1) / nd; // contiguous blocks in TFAM order
    for (int d = 0; d < nd; d++) {
        if (std::min(per, impl->nind - d * per) < 1) break;
        impl->shards.emplace_back();
        Impl::Shard &s = impl->shards.back();
        s.device = devs[d];
        s.ind_begin = d * per;
        s.nind = std::min(per, impl->nind - s.ind_begin);
        check(garlic_ctx_create(s.device, nullptr, &s.ctx), "garlic_ctx_create");
        check(garlic_panel_create(s.ctx, nchr, impl->chr_nloci.data(), s.nind, &s.panel), "garlic_panel_create");
        check(garlic_panel_set_map(s.panel, pos.data(), gpos.data(), cs.data(), ce.data()), "garlic_panel_set_map");
        check(garlic_panel_set_freq(s.panel, freq.data()), "garlic_panel_set_freq");
    }
    // genotype rows are separate allocations in HapData: stage a slab of SNP rows at a time
    const int64_t slab = std::max<int64_t>(1, ((int64_t)64 << 20) / (2 * (int64_t)impl->nind));
    std::vector<int16_t> stage;
    std::vector<uint8_t> stage2, stage_glc;
    std::vector<double> stage_gl;
    std::vector<uint8_t> stage_fc;
    impl->have_phase = true;
    for (auto h : *haps) impl->have_phase = impl->have_phase && h->firstCopy;
    o = 0;
    for (int c = 0; c < nchr; c++) {
        const HapData *h = haps->at(c);
        for (int l0 = 0; l0 < h->nloci; l0 += (int)slab) {
            const int rows = (int)std::min<int64_t>(slab, h->nloci - l0);
            const size_t row_bytes = ((size_t)impl->nind + 3) / 4;
            if (h->packed) {
                stage2.resize((size_t)rows * row_bytes);
                for (int r = 0; r < rows; r++) memcpy(&stage2[(size_t)r * row_bytes], h->packed[l0 + r], row_bytes);
            } else {
                stage.resize((size_t)rows * impl->nind);
                for (int r = 0; r < rows; r++)
                    memcpy(&stage[(size_t)r * impl->nind], h->data[l0 + r], sizeof(short) * impl->nind);
            }
            const GenoLikeData *gld = USE_GL ? gls->at(c) : nullptr;
            if (gld && gld->codes) {
                stage_glc.resize((size_t)rows * impl->nind);
                for (int r = 0; r < rows; r++) memcpy(&stage_glc[(size_t)r * impl->nind], gld->codes[l0 + r], impl->nind);
            } else if (gld) {
                stage_gl.resize((size_t)rows * impl->nind);
                for (int r = 0; r < rows; r++)
                    memcpy(&stage_gl[(size_t)r * impl->nind], gld->data[l0 + r], sizeof(double) * impl->nind);
            }
            if (impl->have_phase) {
                stage_fc.resize((size_t)rows * impl->nind);
                for (int r = 0; r < rows; r++)
                    for (int i = 0; i < impl->nind; i++)
                        stage_fc[(size_t)r * impl->nind + i] = h->firstCopy[l0 + r][i];
            }
            for (auto &s : impl->shards) {
                if (h->packed)
                    check(garlic_panel_set_genotypes_2bit(s.panel, stage2.data(), (int64_t)row_bytes, s.ind_begin, o + l0,
                                                          rows, GARLIC_HOST), "garlic_panel_set_genotypes_2bit");
                else
                    check(garlic_panel_set_genotypes(s.panel, stage.data() + s.ind_begin, impl->nind, o + l0, rows,
                                                     GARLIC_HOST), "garlic_panel_set_genotypes");
                if (impl->have_phase)
                    check(garlic_panel_set_phase(s.panel, stage_fc.data() + s.ind_begin, impl->nind, o + l0, rows,
                                                 GARLIC_HOST), "garlic_panel_set_phase");
                if (gld && gld->codes)
                    check(garlic_panel_set_gl_codes(s.panel, stage_glc.data() + s.ind_begin, impl->nind, o + l0, rows,
                                                    gld->values, gld->nvalues, GARLIC_HOST), "garlic_panel_set_gl_codes");
                else if (gld)
                    check(garlic_panel_set_gl(s.panel, stage_gl.data() + s.ind_begin, impl->nind, o + l0, rows,
                                              GARLIC_HOST), "garlic_panel_set_gl");
            }
        }
        o += h->nloci;
    }
}

LodEngine::~LodEngine() { release(); }

std::vector<WinData *> *LodEngine::lodWindows(int winsize, double error, int MAX_GAP)
{
    return run(false, winsize, error, MAX_GAP, 0, 0.0);
}

std::vector<WinData *> *LodEngine::wlodWindows(std::vector<LDData *> *lds, int winsize, double error, int MAX_GAP,
                                               int M, double mu)
{
    // LDData rows are separate allocations: flatten to [locus][winsize] and hand to every device
    int64_t nloci = 0;
    for (int n : impl->chr_nloci) nloci += n;
    std::vector<double> flat((size_t)nloci * winsize);
    int64_t o = 0;
    for (size_t c = 0; c < lds->size(); c++)
        for (int l = 0; l < lds->at(c)->nloci; l++, o++)
            memcpy(&flat[(size_t)o * winsize], lds->at(c)->LD[l], sizeof(double) * winsize);
    for (auto &s : impl->shards) check(garlic_panel_set_ld(s.panel, winsize, flat.data(), GARLIC_HOST), "garlic_panel_set_ld");
    return run(true, winsize, error, MAX_GAP, M, mu);
}

std::vector<WinData *> *LodEngine::wlodWindowsResident(int winsize, double error, int MAX_GAP, int M, double mu)
{
    return run(true, winsize, error, MAX_GAP, M, mu);
}

DoubleData *LodEngine::lodFeed(int winsize, double error, int MAX_GAP, int step, bool weighted, int M, double mu,
                               const std::vector<int> *kdeSubsample)
{
    std::cerr << "Calculating LOD scores with winsize " << winsize << " (thinned on the device, step " << step << ").\n";
    const int nchr = (int)impl->chr_nloci.size();
    const size_t ns = impl->shards.size();
    // --kde-subsample (convertSubsetWinData2DoubleData, garlic-data.cpp:2071-2150): panel-wide indices in
    // increasing order (what gsl_ran_choose returns), so that the shards' feeds -- each over its own part
    // of the list -- merge per chromosome in shard order exactly as for the whole panel
    const bool subset = kdeSubsample && !kdeSubsample->empty();
    if (subset)
        for (size_t i = 1; i < kdeSubsample->size(); i++)
            if ((*kdeSubsample)[i] <= (*kdeSubsample)[i - 1]) fail("KDE subsample must be in increasing order");
    std::vector<std::vector<double>> feeds(ns);
    std::vector<std::vector<int64_t>> per_chr(ns, std::vector<int64_t>(nchr, 0));
    std::vector<std::string> errors(ns);
    std::vector<std::thread> th;
    for (size_t k = 0; k < ns; k++)
        th.emplace_back([&, k] {
            auto &s = impl->shards[k];
            std::vector<int32_t> mine;
            if (subset) {
                for (int g : *kdeSubsample)
                    if (g >= s.ind_begin && g < s.ind_begin + s.nind) mine.push_back(g - s.ind_begin);
                if (mine.empty()) { feeds[k].clear(); return; }      // no listed individual lives here
            }
            const int64_t rows = subset ? (int64_t)mine.size() : s.nind;
            int64_t cap = 0, n = 0;
            for (int c = 0; c < nchr; c++) cap += ((int64_t)impl->chr_nloci[c] + step - 1) / step * rows;
            feeds[k].resize((size_t)std::max<int64_t>(cap, 1));
            if (garlic_lod_feed_subset(s.panel, winsize, error, MAX_GAP, impl->use_gl, weighted, M, mu, step,
                                       subset ? mine.data() : nullptr, (int32_t)mine.size(), feeds[k].data(), cap, &n,
                                       per_chr[k].data()) != GARLIC_OK)
                errors[k] = garlic_hip_last_error();
            else
                feeds[k].resize((size_t)n);
        });
    for (auto &t : th) t.join();
    for (auto &e : errors)
        if (!e.empty()) fail("garlic_lod_feed: " + e);
    // merge in the reference's order: chromosome -> individual (= shard order) -> locus
    int64_t total = 0;
    for (auto &f : feeds) total += (int64_t)f.size();
    if (total > 0x7fffffff) fail("KDE feed has more than 2^31 values: DoubleData::size is an int");
    DoubleData *d = new DoubleData;
    d->size = (int)total;
    d->data = new double[total > 0 ? total : 1];
    std::vector<int64_t> off(ns, 0);
    int64_t o = 0;
    for (int c = 0; c < nchr; c++)
        for (size_t k = 0; k < ns; k++) {
            memcpy(d->data + o, feeds[k].data() + off[k], sizeof(double) * (size_t)per_chr[k][c]);
            o += per_chr[k][c];
            off[k] += per_chr[k][c];
        }
    return d;
}

// ---- ROH calls
std::vector<ROHData *> *initROHData(IndData *indData)
{
    auto *v = new std::vector<ROHData *>;
    for (int ind = 0; ind < indData->nind; ind++) v->push_back(new ROHData);
    return v;
}

void releaseROHData(std::vector<ROHData *> *rohDataByInd)
{
    if (!rohDataByInd) return;
    for (ROHData *r : *rohDataByInd) delete r;
    rohDataByInd->clear();
    delete rohDataByInd;
}

ROHLength *initROHLength(int size, std::string pop)
{
    ROHLength *r = new ROHLength;
    r->pop = pop;
    r->length = new double[size > 0 ? size : 1];
    r->size = size;
    return r;
}

void releaseROHLength(ROHLength *rohLength)
{
    if (!rohLength) return;
    delete[] rohLength->length;
    delete rohLength;
}

// garlic-roh.cpp:574-648
void writeROHData(const std::string &outfile, std::vector<ROHData *> *rohDataByInd, std::vector<MapData *> *mapDataByChr,
                  const std::vector<double> &bounds, const std::string &popName, const std::string &version, bool CM)
{
    static const char *colors[9] = {"228,26,28", "77,175,74", "55,126,184", "152,78,163", "255,127,0",
                                    "255,255,51", "166,86,40", "247,129,191", "153,153,153"};
    std::ofstream out(outfile.c_str());
    if (out.fail()) fail("Failed to open " + outfile);
    for (size_t ind = 0; ind < rohDataByInd->size(); ind++) {
        const ROHData *rohData = rohDataByInd->at(ind);
        out << "track name=\"Ind: " + rohData->indID + " Pop:" + popName + " ROH\" description=\"Ind: " + rohData->indID +
                   " Pop:" + popName + " ROH from GARLIC v" + version + "\" visibility=2 itemRgb=\"On\"\n";
        for (size_t roh = 0; roh < rohData->chr.size(); roh++) {
            const double size = rohData->length[roh];
            // the first boundary the size lies below names the class (A, B, ..); past the last one: the next letter
            size_t i = 0;
            while (i < bounds.size() && !(size < bounds[i])) i++;
            const char sizeClass = (char)('A' + i);
            const char *color = colors[i <= 8 ? i : 8];
            std::string chr = mapDataByChr->at((size_t)rohData->chr[roh])->chr;
            if (chr[0] != 'c' && chr[0] != 'C') chr = "chr" + chr;
            out << chr << "\t" << int(rohData->start[roh]) << "\t" << int(rohData->stop[roh]) << "\t" << sizeClass << "\t";
            if (CM) out << size;
            else out << int(size);
            out << "\t.\t0\t0\t" << color << std::endl;
        }
    }
    out.close();
    std::cerr << "ROH calls: " << outfile << "\n";
}

// assembleROHWindows, garlic-roh.cpp:409-545: the device returns every segment as (individual, chromosome, first SNP,
// last SNP) in the reference's order; positions, sizes and the pooled length list are filled in here as :470-520 do
// (size in cM from geneticPos with CM, else stop - start + 1 in bp; start / stop are the SNPs' physical positions)
std::vector<ROHData *> *LodEngine::assembleROHWindows(IndData *indData, double lodScoreCutoff, ROHLength **rohLength,
                                                      int winSize, double error, int MAX_GAP, double OVERLAP_FRAC, bool CM,
                                                      bool weighted, int M, double mu)
{
    if (!indData || indData->nind != impl->nind) fail("assembleROHWindows: IndData does not match the panel");
    std::cerr << "Assembling ROH windows on the device (winsize " << winSize << ").\n";
    const size_t ns = impl->shards.size();
    std::vector<std::vector<garlic_roh_segment>> segs(ns);
    std::vector<std::string> errors(ns);
    std::vector<std::thread> th;
    for (size_t k = 0; k < ns; k++)
        th.emplace_back([&, k] {
            auto &s = impl->shards[k];
            int64_t n = 0;
            // a first guess of room for 64 segments per individual; the call says how many there are
            segs[k].resize((size_t)std::max<int64_t>(1024, 64 * (int64_t)s.nind));
            for (int attempt = 0; attempt < 2; attempt++) {
                if (garlic_roh_segments(s.panel, winSize, error, MAX_GAP, impl->use_gl, weighted, M, mu, lodScoreCutoff, OVERLAP_FRAC,
                                        segs[k].data(), (int64_t)segs[k].size(), &n) != GARLIC_OK) {
                    errors[k] = garlic_hip_last_error();
                    return;
                }
                if (n <= (int64_t)segs[k].size()) break;
                segs[k].resize((size_t)n);
            }
            segs[k].resize((size_t)n);
        });
    for (auto &t : th) t.join();
    for (auto &e : errors)
        if (!e.empty()) fail("garlic_roh_segments: " + e);
    std::vector<ROHData *> *rohDataByInd = initROHData(indData);
    std::vector<double> lengths;
    for (size_t k = 0; k < ns; k++) {           // shards hold the individuals in order, each list is sorted by individual
        const auto &s = impl->shards[k];
        for (const garlic_roh_segment &g : segs[k]) {
            const MapData *map = impl->maps->at((size_t)g.chr);
            ROHData *r = rohDataByInd->at((size_t)(s.ind_begin + g.ind));
            const int winStart = map->physicalPos[g.start], winStop = map->physicalPos[g.stop];
            const double size = CM ? map->geneticPos[g.stop] - map->geneticPos[g.start] : winStop - winStart + 1;
            lengths.push_back(size);
            r->length.push_back(size);
            r->chr.push_back(g.chr);
            r->start.push_back(winStart);
            r->stop.push_back(winStop);
        }
    }
    for (int ind = 0; ind < indData->nind; ind++) rohDataByInd->at((size_t)ind)->indID = indData->indID[ind];
    ROHLength *rl = initROHLength((int)lengths.size(), indData->pop);
    for (size_t i = 0; i < lengths.size(); i++) rl->length[i] = lengths[i];
    if (rohLength) *rohLength = rl;
    else releaseROHLength(rl);
    return rohDataByInd;
}

// The callers that sweep window sizes on one data set -- exploreWinsizes (garlic-roh.cpp:726-751), selectWinsize
// (:798-837), selectWinsizeFromList (:881-920) -- through garlic_lod_feed_multi: unweighted --error scores, the thinning
// step of a size is the size itself (convertWinData2DoubleData(.., winsize), :735,743,817,900) unless `steps` says otherwise.
std::vector<DoubleData *> LodEngine::lodFeedMulti(const std::vector<int> &winsizes, double error, int MAX_GAP,
                                                  const std::vector<int> *steps, const std::vector<int> *kdeSubsample)
{
    const int nchr = (int)impl->chr_nloci.size();
    const size_t ns = impl->shards.size(), nw = winsizes.size();
    if (nw == 0) return {};
    if (steps && steps->size() != nw) fail("lodFeedMulti: one thinning step per window size");
    std::cerr << "Calculating LOD scores with winsizes";
    for (int W : winsizes) std::cerr << " " << W;
    std::cerr << " (thinned on the device).\n";
    const bool subset = kdeSubsample && !kdeSubsample->empty();
    if (subset)
        for (size_t i = 1; i < kdeSubsample->size(); i++)
            if ((*kdeSubsample)[i] <= (*kdeSubsample)[i - 1]) fail("KDE subsample must be in increasing order");
    std::vector<int32_t> W32(winsizes.begin(), winsizes.end()), S32(nw);
    for (size_t i = 0; i < nw; i++) S32[i] = steps ? (*steps)[i] : winsizes[i];
    // feeds[shard][size], per_chr[shard][size * nchr + c]
    std::vector<std::vector<std::vector<double>>> feeds(ns, std::vector<std::vector<double>>(nw));
    std::vector<std::vector<int64_t>> per_chr(ns, std::vector<int64_t>(nw * (size_t)nchr, 0));
    std::vector<std::string> errors(ns);
    std::vector<std::thread> th;
    for (size_t k = 0; k < ns; k++)
        th.emplace_back([&, k] {
            auto &s = impl->shards[k];
            std::vector<int32_t> mine;
            if (subset) {
                for (int g : *kdeSubsample)
                    if (g >= s.ind_begin && g < s.ind_begin + s.nind) mine.push_back(g - s.ind_begin);
                if (mine.empty()) return;      // no listed individual lives here
            }
            const int64_t rows = subset ? (int64_t)mine.size() : s.nind;
            std::vector<double *> ptrs(nw);
            std::vector<int64_t> cap(nw, 0), n(nw, 0);
            for (size_t i = 0; i < nw; i++) {
                for (int c = 0; c < nchr; c++) cap[i] += ((int64_t)impl->chr_nloci[c] + S32[i] - 1) / S32[i] * rows;
                feeds[k][i].resize((size_t)std::max<int64_t>(cap[i], 1));
                ptrs[i] = feeds[k][i].data();
            }
            if (garlic_lod_feed_multi(s.panel, W32.data(), S32.data(), (int32_t)nw, error, MAX_GAP, subset ? mine.data() : nullptr,
                                      (int32_t)mine.size(), ptrs.data(), cap.data(), n.data(), per_chr[k].data()) != GARLIC_OK)
                errors[k] = garlic_hip_last_error();
            else
                for (size_t i = 0; i < nw; i++) feeds[k][i].resize((size_t)n[i]);
        });
    for (auto &t : th) t.join();
    for (auto &e : errors)
        if (!e.empty()) fail("garlic_lod_feed_multi: " + e);
    // merge per size in the reference's order: chromosome -> individual (= shard order) -> locus
    std::vector<DoubleData *> out(nw, nullptr);
    for (size_t i = 0; i < nw; i++) {
        int64_t total = 0;
        for (size_t k = 0; k < ns; k++) total += (int64_t)feeds[k][i].size();
        if (total > 0x7fffffff) fail("KDE feed has more than 2^31 values: DoubleData::size is an int");
        DoubleData *d = new DoubleData;
        d->size = (int)total;
        d->data = new double[total > 0 ? total : 1];
        std::vector<int64_t> off(ns, 0);
        int64_t o = 0;
        for (int c = 0; c < nchr; c++)
            for (size_t k = 0; k < ns; k++) {
                const int64_t m = feeds[k][i].empty() ? 0 : per_chr[k][i * (size_t)nchr + c];
                memcpy(d->data + o, feeds[k][i].data() + off[k], sizeof(double) * (size_t)m);
                o += m;
                off[k] += m;
            }
        out[i] = d;
    }
    return out;
}

std::vector<LDData *> *LodEngine::ldWeights(int winsize, const std::vector<int> &subsample, bool want_host,
                                            bool phased)
{
    if (phased && !impl->have_phase) fail("--phased: the genotypes were loaded without phase (HapData::firstCopy)");
    const int32_t ph = phased ? 1 : 0;
    std::cerr << "Calculating LD weights with winsize " << winsize << ".\n";
    int64_t nloci = 0;
    for (int n : impl->chr_nloci) nloci += n;
    // every shard counts over its own individuals; the counts are integers, so their sum over
    // shards is exact whatever the order (with one process per GPU this is an RCCL all-reduce)
    std::vector<int32_t> loc((size_t)nloci * 2, 0), pair((size_t)nloci * winsize * 2, 0);
    const size_t ns = impl->shards.size();
    std::vector<std::string> errors(ns);
    std::mutex sum_lock;
    auto count_shard = [&](size_t k) {   // one host thread per GPU, as for the scores
        auto &s = impl->shards[k];
        std::vector<int32_t> sub;
        for (int g : subsample)
            if (g >= s.ind_begin && g < s.ind_begin + s.nind) sub.push_back(g - s.ind_begin);
        std::vector<int32_t> l1, p1;
        std::vector<int32_t> &lo = ns == 1 ? loc : l1, &pa = ns == 1 ? pair : p1;
        lo.resize(loc.size()); pa.resize(pair.size());
        // subsample given: the shard's part of it, possibly empty (a non-NULL pointer with n_sub = 0 means
        // "none of them": pair counts zero, the homFreq counts still over every individual)
        static const int32_t no_index = 0;
        const int32_t *sub_ptr = subsample.empty() ? nullptr : (sub.empty() ? &no_index : sub.data());
        if (garlic_ld_counts(s.panel, winsize, ph, sub_ptr, (int32_t)sub.size(), lo.data(), pa.data(), GARLIC_HOST) !=
            GARLIC_OK) {
            errors[k] = garlic_hip_last_error();
            return;
        }
        if (ns == 1) return;
        std::lock_guard<std::mutex> hold(sum_lock);
        for (size_t i = 0; i < loc.size(); i++) loc[i] += lo[i];
        for (size_t i = 0; i < pair.size(); i++) pair[i] += pa[i];
    };
    {
        std::vector<std::thread> th;
        for (size_t k = 0; k < ns; k++) th.emplace_back(count_shard, k);
        for (auto &t : th) t.join();
    }
    for (auto &e : errors)
        if (!e.empty()) fail("garlic_ld_counts: " + e);
    std::vector<double> flat;
    if (want_host) flat.resize((size_t)nloci * winsize);
    {
        std::vector<std::thread> th;
        for (size_t k = 0; k < ns; k++)
            th.emplace_back([&, k] {
                if (garlic_ld_finish(impl->shards[k].panel, winsize, ph, loc.data(), pair.data(),
                                     (want_host && k == 0) ? flat.data() : nullptr, GARLIC_HOST) != GARLIC_OK)
                    errors[k] = garlic_hip_last_error();
            });
        for (auto &t : th) t.join();
    }
    for (auto &e : errors)
        if (!e.empty()) fail("garlic_ld_finish: " + e);
    if (!want_host) return nullptr;
    std::vector<LDData *> *out = new std::vector<LDData *>;
    int64_t o = 0;
    for (int n : impl->chr_nloci) {
        LDData *d = initLDData(n, winsize);
        for (int l = 0; l < n; l++, o++) memcpy(d->LD[l], &flat[(size_t)o * winsize], sizeof(double) * winsize);
        out->push_back(d);
    }
    return out;
}

std::vector<WinData *> *LodEngine::run(bool weighted, int winsize, double error, int MAX_GAP, int M, double mu)
{
    std::cerr << "Calculating LOD scores with winsize " << winsize << ".\n";
    std::vector<WinData *> *win = initWinData(impl->maps, impl->nind);
    const int nchr = (int)impl->chr_nloci.size();
    std::vector<std::string> errors(impl->shards.size());
    std::vector<std::thread> th;
    for (size_t k = 0; k < impl->shards.size(); k++) {
        th.emplace_back([&, k] { // one host thread per GPU; no collective, just a gather of rows
            auto &s = impl->shards[k];
            std::vector<int64_t> base(nchr), pitch(nchr);
            int64_t total = 0;
            // individuals come back in chunks so the host never stages more than ~1 GiB
            int64_t per_ind = 0;
            for (int c = 0; c < nchr; c++) per_ind += impl->chr_nloci[c];
            const int chunk = (int)std::max<int64_t>(1, std::min<int64_t>(s.nind, ((int64_t)1 << 27) / std::max<int64_t>(1, per_ind)));
            std::vector<double> buf;
            for (int i0 = 0; i0 < s.nind; i0 += chunk) {
                const int n = std::min(chunk, s.nind - i0);
                if (garlic_lod_out_layout(s.panel, 1, n, base.data(), pitch.data(), &total) != GARLIC_OK ||
                    (buf.resize((size_t)total), false) ||
                    (weighted ? garlic_wlod_windows(s.panel, winsize, error, MAX_GAP, impl->use_gl, M, mu, i0, n, 1,
                                                    buf.data(), GARLIC_HOST)
                              : garlic_lod_windows(s.panel, winsize, error, MAX_GAP, impl->use_gl, i0, n, 1,
                                                   buf.data(), GARLIC_HOST)) != GARLIC_OK) {
                    errors[k] = garlic_hip_last_error();
                    return;
                }
                for (int c = 0; c < nchr; c++)
                    for (int i = 0; i < n; i++)
                        memcpy(win->at(c)->data[s.ind_begin + i0 + i], &buf[base[c] + (int64_t)i * pitch[c]],
                               sizeof(double) * impl->chr_nloci[c]);
            }
        });
    }
    for (auto &t : th) t.join();
    for (auto &e : errors)
        if (!e.empty()) { releaseWinData(win); fail("garlic_lod_windows: " + e); }
    return win;
}

std::vector<int> drawLdSubsample(int nind, int ldSubsample, unsigned long long seed)
{
    std::vector<int> out;
    if (ldSubsample >= nind || ldSubsample <= 0) return out;   // all (garlic-data.cpp:351-356)
    // selection sampling: ldSubsample of nind, each subset equally likely, indices ascending --
    // the contract of gsl_ran_choose; the stream itself cannot match a time-seeded GSL generator
    std::mt19937_64 rng(seed ? seed : (unsigned long long)time(nullptr));
    int need = ldSubsample;
    for (int i = 0; i < nind && need > 0; i++) {
        const double u = std::generate_canonical<double, 53>(rng);
        if ((double)(nind - i) * u < (double)need) { out.push_back(i); need--; }
    }
    return out;
}

std::vector<int> drawKdeSubsample(int nind, int kdeSubsample, unsigned long long seed)
{   // garlic-data.cpp:2081-2096: everyone when the subsample is not smaller than the panel (here: empty list)
    return drawLdSubsample(nind, kdeSubsample, seed ? seed ^ 0x6B64655F73756273ull : 0);
}

std::vector<LDData *> *calcLDData(std::vector<HapData *> *haps, std::vector<FreqData *> *freqs,
                                  std::vector<MapData *> *maps, std::vector<GenoFreqData *> * /*genoFreq*/,
                                  centromere *centro, int winsize, int /*MAX_GAP*/, bool PHASED,
                                  int /*numThreads*/, int ldSubsample)
{
    LodEngine engine(haps, freqs, maps, nullptr, centro, false, g_options.devices);
    return engine.ldWeights(winsize, drawLdSubsample(haps->at(0)->nind, ldSubsample, g_options.ld_seed), true,
                            PHASED);
}

std::vector<WinData *> *calcLODWindows(std::vector<HapData *> *haps, std::vector<FreqData *> *freqs,
                                       std::vector<MapData *> *maps, std::vector<GenoLikeData *> *gls,
                                       centromere *centro, int winsize, double error, int MAX_GAP, bool USE_GL)
{
    // GLDataByChr may be an uninitialised pointer when !USE_GL (garlic-roh.cpp:713,720): never touched then
    LodEngine engine(haps, freqs, maps, USE_GL ? gls : nullptr, centro, USE_GL, g_options.devices);
    return engine.lodWindows(winsize, error, MAX_GAP);
}

std::vector<WinData *> *calcwLODWindows(std::vector<HapData *> *haps, std::vector<FreqData *> *freqs,
                                        std::vector<MapData *> *maps, std::vector<GenoLikeData *> *gls,
                                        std::vector<LDData *> *lds, centromere *centro, int winsize, double error,
                                        int MAX_GAP, bool USE_GL, int M, double mu, int /*numThreads*/)
{
    // numThreads only partitions loci in the reference (garlic-data.cpp:538); the result does not depend on it
    LodEngine engine(haps, freqs, maps, USE_GL ? gls : nullptr, centro, USE_GL, g_options.devices);
    return engine.wlodWindows(lds, winsize, error, MAX_GAP, M, mu);
}

// ------------------------------------------------------------------------- consumers
DoubleData *convertWinData2DoubleData(std::vector<WinData *> *wins, int step)
{
    // garlic-data.cpp:2026-2069: chr -> ind -> every step-th locus; MISSING and NaN are dropped
    int size = 0;
    for (auto w : *wins)
        for (int i = 0; i < w->nind; i++)
            for (int l = 0; l < w->nloci; l += step) {
                const double x = w->data[i][l];
                if (x != MISSING && !std::isnan(x)) size++;
            }
    DoubleData *d = new DoubleData;
    d->size = size;
    d->data = new double[size > 0 ? size : 1];
    int k = 0;
    for (auto w : *wins)
        for (int i = 0; i < w->nind; i++)
            for (int l = 0; l < w->nloci; l += step) {
                const double x = w->data[i][l];
                if (x != MISSING && !std::isnan(x)) d->data[k++] = x;
            }
    return d;
}

DoubleData *convertSubsetWinData2DoubleData(std::vector<WinData *> *wins, const std::vector<int> &randInd, int step)
{
    // garlic-data.cpp:2071-2150 with the draw supplied: chr -> randInd[0..] -> every step-th locus
    int size = 0;
    for (auto w : *wins)
        for (int i : randInd) {
            if (i < 0 || i >= w->nind) { std::cerr << "ERROR: KDE subsample index " << i << " outside the panel.\n"; throw 0; }
            for (int l = 0; l < w->nloci; l += step) {
                const double x = w->data[i][l];
                if (x != MISSING && !std::isnan(x)) size++;
            }
        }
    DoubleData *d = new DoubleData;
    d->size = size;
    d->data = new double[size > 0 ? size : 1];
    int k = 0;
    for (auto w : *wins)
        for (int i : randInd)
            for (int l = 0; l < w->nloci; l += step) {
                const double x = w->data[i][l];
                if (x != MISSING && !std::isnan(x)) d->data[k++] = x;
            }
    return d;
}

// one gzip member holding `text` (members may simply follow each other in a .gz file)
static std::string gzip_member(const std::string &text)
{
    z_stream z;
    memset(&z, 0, sizeof z);
    if (deflateInit2(&z, Z_DEFAULT_COMPRESSION, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw -1;
    std::string out;
    out.resize(deflateBound(&z, (uLong)text.size()) + 64);
    z.next_in = reinterpret_cast<Bytef *>(const_cast<char *>(text.data()));
    z.avail_in = (uInt)text.size();
    z.next_out = reinterpret_cast<Bytef *>(&out[0]);
    z.avail_out = (uInt)out.size();
    const int rc = deflate(&z, Z_FINISH);
    out.resize(out.size() - z.avail_out);
    deflateEnd(&z);
    if (rc != Z_STREAM_END) throw -1;
    return out;
}

// The reference's text (garlic-data.cpp:1722-1745: one line per individual, "NA" for MISSING, operator<<'s six
// significant digits = printf's %g), written the way a GPU-sized panel needs it: the individuals' lines are formatted and
// compressed on all host cores, a few lines per gzip member, and appended in order (formatted through one
// ostringstream and one gz stream, 20 M values took 8 s).
void writeWinData(std::vector<WinData *> *wins, IndData *indData, std::vector<MapData *> *maps,
                  const std::string &outfile)
{
    const unsigned nthreads = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    for (size_t c = 0; c < maps->size(); c++) {
        const std::string path = outfile + "." + indData->pop + "." + maps->at(c)->chr + ".raw.lod.windows.gz";
        FILE *f = fopen(path.c_str(), "wb");
        if (!f) { std::cerr << "ERROR: Failed to open " << path << " for writing.\n"; throw -1; }
        const WinData *w = wins->at(c);
        // rows per member: ~4 MB of text each; a wave of members = one per thread
        const int per = std::max(1, (int)(((size_t)4 << 20) / ((size_t)std::max(1, w->nloci) * 8 + 1)));
        for (int i0 = 0; i0 < w->nind; i0 += per * (int)nthreads) {
            const int ntasks = std::min<int>((int)nthreads, (w->nind - i0 + per - 1) / per);
            std::vector<std::string> members((size_t)ntasks);
            std::vector<std::thread> pool;
            std::atomic<bool> failed{false};   // set by the compression threads
            for (int t = 0; t < ntasks; t++)
                pool.emplace_back([&, t]() {
                    std::string text;
                    text.reserve((size_t)per * ((size_t)w->nloci * 9 + 1));
                    char buf[40];
                    const int a = i0 + t * per, b = std::min(w->nind, a + per);
                    for (int i = a; i < b; i++) {
                        for (int l = 0; l < w->nloci; l++) {
                            if (w->data[i][l] == MISSING) text += "NA";               // garlic-data.cpp:1736
                            else text.append(buf, (size_t)snprintf(buf, sizeof buf, "%g", w->data[i][l]));
                            if (l < w->nloci - 1) text += ' ';
                        }
                        text += '\n';
                    }
                    try { members[(size_t)t] = gzip_member(text); } catch (...) { failed = true; }
                });
            for (auto &th : pool) th.join();
            if (failed) { fclose(f); std::cerr << "ERROR: Failed to compress " << path << "\n"; throw -1; }
            for (const std::string &m : members)
                if (fwrite(m.data(), 1, m.size(), f) != m.size()) { fclose(f); std::cerr << "ERROR: Failed to write " << path << "\n"; throw -1; }
        }
        if (w->nind == 0) { const std::string m = gzip_member(""); fwrite(m.data(), 1, m.size(), f); }
        fclose(f);
        std::cerr << "Wrote " << path << "\n";
    }
}

} // namespace garlic_host
