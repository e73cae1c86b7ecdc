// CPU unit checks of the host adapter's compact containers (no GPU needed): the packed genotype rows of
// the cache and the dictionary-coded likelihoods must describe exactly what the reference-shaped
// containers hold, before and after the site filter.
#include "../../garlic_amd/host/garlic_host.hpp"

#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <sstream>
#include <fstream>

using namespace garlic_host;

#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) { std::cerr << "FAILED " << __LINE__ << ": " #cond "\n"; return 1; } \
    } while (0)

static bool same_genotypes(std::vector<HapData *> *a, std::vector<HapData *> *b)
{
    if (a->size() != b->size()) return false;
    for (size_t c = 0; c < a->size(); c++) {
        if (a->at(c)->nloci != b->at(c)->nloci || a->at(c)->nind != b->at(c)->nind) return false;
        for (int l = 0; l < a->at(c)->nloci; l++)
            for (int i = 0; i < a->at(c)->nind; i++)
                if (genotypeAt(a->at(c), l, i) != genotypeAt(b->at(c), l, i)) return false;
    }
    return true;
}

int main(int argc, char **argv)
{
    if (argc != 5) { std::cerr << "usage: host_unit tped tgls gl_type tmpdir\n"; return 2; }
    const std::string tped = argv[1], tgls = argv[2], gl_type = argv[3], tmp = argv[4];
    try {
        int nl = 0, ni = 0, nl2 = 0, ni2 = 0, nl3 = 0, ni3 = 0;
        std::vector<HapData *> *h, *hp, *hs;
        std::vector<MapData *> *m, *mp, *ms;
        std::vector<FreqData *> *f, *fp, *fs;
        loadTPEDData(tped, nl, ni, &h, &m, &f, '0', /*PHASED=*/true);
        CHECK(h->at(0)->data && h->at(0)->firstCopy && !h->at(0)->packed);
        const std::string cache = tmp + "/unit.g2b";
        writeGenotypeCache(cache, h, m, f);
        loadGenotypeCache(cache, nl2, ni2, &hp, &mp, &fp, /*keepPacked=*/true);
        loadGenotypeCache(cache, nl3, ni3, &hs, &ms, &fs, /*keepPacked=*/false);
        CHECK(nl2 == nl && ni2 == ni && nl3 == nl && ni3 == ni);
        CHECK(hp->at(0)->packed && !hp->at(0)->data && hs->at(0)->data && !hs->at(0)->packed);
        CHECK(same_genotypes(h, hp) && same_genotypes(h, hs));
        for (size_t c = 0; c < h->size(); c++)           // phase bits and frequencies survive the cache
            for (int l = 0; l < h->at(c)->nloci; l++) {
                CHECK(memcmp(&f->at(c)->freq[l], &fp->at(c)->freq[l], sizeof(double)) == 0);
                for (int i = 0; i < ni; i++) CHECK(h->at(c)->firstCopy[l][i] == hp->at(c)->firstCopy[l][i]);
            }
        // a cache written from packed rows is byte-identical
        const std::string cache2 = tmp + "/unit2.g2b";
        writeGenotypeCache(cache2, hp, mp, fp);
        {
            FILE *a = fopen(cache.c_str(), "rb"), *b = fopen(cache2.c_str(), "rb");
            CHECK(a && b);
            int ca, cb;
            do { ca = fgetc(a); cb = fgetc(b); CHECK(ca == cb); } while (ca != EOF);
            fclose(a); fclose(b);
        }
        // likelihoods: dictionary codes describe the same doubles
        std::vector<GenoLikeData *> *g = readTGLSData(tgls, nl, ni, m, gl_type, false);
        std::vector<GenoLikeData *> *gc = readTGLSData(tgls, nl, ni, mp, gl_type, true);
        for (size_t c = 0; c < g->size(); c++) {
            CHECK(g->at(c)->data && !g->at(c)->codes && gc->at(c)->codes && !gc->at(c)->data);
            CHECK(gc->at(c)->nvalues >= 1 && gc->at(c)->nvalues <= 256);
            for (int l = 0; l < g->at(c)->nloci; l++)
                for (int i = 0; i < ni; i++) {
                    const double x = likelihoodAt(g->at(c), l, i), y = likelihoodAt(gc->at(c), l, i);
                    CHECK(memcmp(&x, &y, sizeof x) == 0);
                }
        }
        // the site filter keeps the same sites and rows in both forms
        const int k1 = filterMonomorphicSites(&m, &h, &f, &g, true);
        const int k2 = filterMonomorphicSites(&mp, &hp, &fp, &gc, true);
        CHECK(k1 == k2 && k1 < nl);
        CHECK(same_genotypes(h, hp));
        for (size_t c = 0; c < g->size(); c++)
            for (int l = 0; l < g->at(c)->nloci; l++)
                for (int i = 0; i < ni; i++) {
                    const double x = likelihoodAt(g->at(c), l, i), y = likelihoodAt(gc->at(c), l, i);
                    CHECK(memcmp(&x, &y, sizeof x) == 0);
                }
        releaseGLData(g); releaseGLData(gc);
        releaseHapData(h); releaseHapData(hp); releaseHapData(hs);
        releaseMapData(m); releaseMapData(mp); releaseMapData(ms);
        releaseFreqData(f); releaseFreqData(fp); releaseFreqData(fs);
        // writeWinData: the text (inflated, members concatenated) is what one ostringstream per line gives --
        // garlic-data.cpp:1722-1745 -- for ordinary, MISSING, NaN, infinite and tiny values; many lines per gzip
        // member (narrow chromosome) and many members (wide one)
        {
            const int nind = 37, sizes[2] = {5, 200000};
            std::vector<MapData *> wm;
            std::vector<WinData *> ww;
            IndData ind;
            ind.pop = "popX";
            ind.nind = nind;
            ind.indID = nullptr;
            uint64_t x = 88172645463325252ull;
            for (int c = 0; c < 2; c++) {
                MapData *m = new MapData;
                m->physicalPos = nullptr; m->geneticPos = nullptr; m->locusName = nullptr; m->allele = nullptr;
                m->nloci = sizes[c];
                m->chr = c == 0 ? "7" : "X";
                wm.push_back(m);
                WinData *w = initWinData(nind, sizes[c]);
                for (int i = 0; i < nind; i++)
                    for (int l = 0; l < sizes[c]; l++) {
                        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
                        double v = ((double)(x >> 11) / 9007199254740992.0 - 0.5) * ((x & 7) == 0 ? 1e-7 : 2000.0);
                        if ((x & 1023) == 1) v = MISSING;
                        if ((x & 1023) == 2) v = std::nan("");
                        if ((x & 1023) == 3) v = -std::nan("");
                        if ((x & 1023) == 4) v = -INFINITY;
                        if ((x & 1023) == 5) v = -0.0;
                        w->data[i][l] = v;
                    }
                ww.push_back(w);
            }
            const std::string base = std::string(argv[4]) + "/rawlod";
            writeWinData(&ww, &ind, &wm, base);
            for (int c = 0; c < 2; c++) {
                const std::string path = base + ".popX." + wm[c]->chr + ".raw.lod.windows.gz";
                gzFile f = gzopen(path.c_str(), "rb");
                CHECK(f != nullptr);
                std::string got;
                static char buf[1 << 16];
                int n;
                while ((n = gzread(f, buf, sizeof buf)) > 0) got.append(buf, (size_t)n);
                gzclose(f);
                std::ostringstream want;
                for (int i = 0; i < nind; i++) {
                    for (int l = 0; l < sizes[c]; l++) {
                        if (ww[c]->data[i][l] == MISSING) want << "NA";
                        else want << ww[c]->data[i][l];
                        if (l < sizes[c] - 1) want << " ";
                    }
                    want << "\n";
                }
                CHECK(got == want.str());
                releaseWinData(ww[c]);
                delete wm[c];
            }
        }
        // writeROHData (garlic-roh.cpp:574-648): track line per individual, size class by the first boundary the size lies
        // below (A, B, ..; past the last: the next letter), bp sizes as integers, cM sizes as the stream prints a double,
        // "chr" put in front of chromosome names that lack it
        {
            IndData ind;
            ind.pop = "POP";
            ind.nind = 2;
            ind.indID = new std::string[2]{"a", "b"};
            std::vector<ROHData *> *roh = initROHData(&ind);
            roh->at(0)->indID = "a";
            roh->at(1)->indID = "b";
            MapData m1, m2;
            m1.chr = "7";
            m2.chr = "chrX";
            std::vector<MapData *> maps{&m1, &m2};
            auto add = [&](int i, int c, int a, int b, double len) {
                roh->at(i)->chr.push_back(c); roh->at(i)->start.push_back(a); roh->at(i)->stop.push_back(b); roh->at(i)->length.push_back(len);
            };
            add(0, 0, 100, 49999, 49900.0);
            add(0, 1, 5, 250004, 250000.0);
            add(1, 0, 7, 100006, 100000.0);
            const std::string bed = tmp + "/unit.roh.bed", bedcm = tmp + "/unit.cm.roh.bed";
            writeROHData(bed, roh, &maps, {50000.0, 200000.0}, "POP", "x", false);
            roh->at(1)->length[0] = 0.125;
            writeROHData(bedcm, roh, &maps, {0.05, 0.2}, "POP", "x", true);
            auto slurp = [](const std::string &p) { std::ifstream f(p); std::stringstream s; s << f.rdbuf(); return s.str(); };
            const std::string t0 = "track name=\"Ind: a Pop:POP ROH\" description=\"Ind: a Pop:POP ROH from GARLIC vx\" visibility=2 itemRgb=\"On\"\n";
            const std::string t1 = "track name=\"Ind: b Pop:POP ROH\" description=\"Ind: b Pop:POP ROH from GARLIC vx\" visibility=2 itemRgb=\"On\"\n";
            CHECK(slurp(bed) == t0 + "chr7\t100\t49999\tA\t49900\t.\t0\t0\t228,26,28\n" + "chrX\t5\t250004\tC\t250000\t.\t0\t0\t55,126,184\n" +
                                t1 + "chr7\t7\t100006\tB\t100000\t.\t0\t0\t77,175,74\n");
            CHECK(slurp(bedcm).find(t1 + "chr7\t7\t100006\tB\t0.125\t.\t0\t0\t77,175,74\n") != std::string::npos);
            releaseROHData(roh);
            delete[] ind.indID;
        }
    } catch (...) {
        std::cerr << "FAILED: exception\n";
        return 1;
    }
    std::cout << "host_unit ok\n";
    return 0;
}
