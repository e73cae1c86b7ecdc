// garlic-lod: GARLIC's Phase I (window LOD scores) as a stand-alone tool on MI355X.
// Keeps the reference's ingest (tped/tfam/tgls/freq/map/centromere) and the Phase-I part of its
// command line (src/garlic-cli.cpp), and writes what GARLIC's next stage consumes:
//   <out>.<pop>.<chr>.raw.lod.windows.gz   with --raw-lod   (src/garlic-data.cpp:1704)
//   <out>.<W>SNPs.lod.f64                  the KDE feed: convertWinData2DoubleData's doubles
//                                          (src/garlic-data.cpp:2026), thinned to every W-th window
//                                          unless --no-kde-thinning
// KDE / ROH assembly / GMM themselves stay in GARLIC (out of scope here).
#include "garlic_host.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

using namespace garlic_host;

namespace {
struct Args {
    std::string tped, tfam, out = "outfile", build = "none", centromere = "none", tgls = "none",
                gl_type = "none", freq_file = "none", map = "none", cache = "none";
    char tped_missing = '0';       // src/garlic-cli.cpp:114
    double error = -1;             // :34 (must be in (0,1) unless TGLS)
    int winsize = 0;               // :38
    std::vector<int> winsize_multi;
    bool auto_winsize = false, weighted = false, raw_lod = false, kde_thinning = true, phased = false;
    bool winsize_stream = false;   // extension: further window sizes from stdin (the loop of selectWinsize, driven by the KDE's owner)
    int auto_winsize_step = 10, max_gap = 200000, M = 7, threads = 1, kde_subsample = 20, gpus = 1;
    std::vector<int> devices;      // --devices; empty = 0 .. gpus-1
    int ld_subsample = 0;          // src/garlic-cli.cpp:137
    int resample = 0;              // src/garlic-cli.cpp:62-64: resamples for the allele frequencies
    unsigned long long resample_seed = 0;   // extension: 0 = time-seeded like the reference
    unsigned long long ld_seed = 0; // extension: 0 = time-seeded like the reference
    unsigned long long kde_seed = 0; // extension: the --kde-subsample draw, 0 = time-seeded like the reference
    double mu = 1e-9, overlap_frac = 0.25;
    bool have_cutoff = false, cm = false;   // --lod-cutoff (src/garlic-cli.cpp:101), --cm (:167)
    double lod_cutoff = 0.0;
    std::vector<double> size_bounds;        // --size-bounds (:107)
};

[[noreturn]] void usage(const char *msg)
{
    std::cerr << "ERROR: " << msg << "\n"
              << "usage: garlic-lod --tped F --tfam F --out P (--build hg18|hg19|hg38 | --centromere F)\n"
                 "         (--error E | --tgls F --gl-type GQ|GL|PL) (--winsize W | --winsize-multi W1 W2 ...)\n"
                 "         [--auto-winsize] [--auto-winsize-step N] [--winsize-stream] [--max-gap N] [--overlap-frac X]\n"
                 "         [--freq-file F] [--tped-missing C] [--raw-lod] [--kde-subsample N] [--kde-seed S] [--no-kde-thinning]\n"
                 "         [--weighted --map F --M N --mu X --ld-subsample N --ld-seed S --threads N]\n"
                 "         [--resample N --resample-seed S] [--gpus N | --devices 0,1,...] [--genotype-cache F]\n"
                 "         [--lod-cutoff X --size-bounds B1 B2 ... [--cm]]   (ROH calls: <out>.roh.bed)\n";
    exit(1);
}

Args parse(int argc, char **argv)
{
    Args a;
    for (int i = 1; i < argc; i++) {
        const std::string f = argv[i];
        auto val = [&]() -> std::string { if (i + 1 >= argc) usage(("missing value for " + f).c_str()); return argv[++i]; };
        if (f == "--tped") a.tped = val();
        else if (f == "--tfam") a.tfam = val();
        else if (f == "--out") a.out = val();
        else if (f == "--build") a.build = val();
        else if (f == "--centromere") a.centromere = val();
        else if (f == "--tgls") a.tgls = val();
        else if (f == "--gl-type") a.gl_type = val();
        else if (f == "--freq-file") a.freq_file = val();
        else if (f == "--map") a.map = val();
        else if (f == "--genotype-cache") a.cache = val();   // extension: binary sidecar of the parsed TPED
        else if (f == "--tped-missing") a.tped_missing = val()[0];
        else if (f == "--error") a.error = atof(val().c_str());
        else if (f == "--winsize") a.winsize = atoi(val().c_str());
        else if (f == "--winsize-multi") {
            while (i + 1 < argc && argv[i + 1][0] != '-') a.winsize_multi.push_back(atoi(argv[++i]));
        }
        else if (f == "--auto-winsize") a.auto_winsize = !a.auto_winsize; // bool flags toggle (param_t.cpp:278)
        else if (f == "--auto-winsize-step") a.auto_winsize_step = atoi(val().c_str());
        else if (f == "--winsize-stream") a.winsize_stream = !a.winsize_stream;
        else if (f == "--max-gap") a.max_gap = atoi(val().c_str());
        else if (f == "--overlap-frac") a.overlap_frac = atof(val().c_str());
        else if (f == "--weighted") a.weighted = !a.weighted;
        else if (f == "--lod-cutoff") { a.lod_cutoff = atof(val().c_str()); a.have_cutoff = true; }
        else if (f == "--cm") a.cm = !a.cm;
        else if (f == "--size-bounds") {
            while (i + 1 < argc && (isdigit((unsigned char)argv[i + 1][0]) || argv[i + 1][0] == '.')) a.size_bounds.push_back(atof(argv[++i]));
        }
        else if (f == "--M") a.M = atoi(val().c_str());
        else if (f == "--mu") a.mu = atof(val().c_str());
        else if (f == "--threads") a.threads = atoi(val().c_str());
        else if (f == "--ld-subsample") a.ld_subsample = atoi(val().c_str());
        else if (f == "--ld-seed") a.ld_seed = strtoull(val().c_str(), nullptr, 10);
        else if (f == "--phased") a.phased = !a.phased;
        else if (f == "--raw-lod") a.raw_lod = !a.raw_lod;
        else if (f == "--kde-subsample") a.kde_subsample = atoi(val().c_str());
        else if (f == "--kde-seed") a.kde_seed = strtoull(val().c_str(), nullptr, 10);
        else if (f == "--resample") a.resample = atoi(val().c_str());
        else if (f == "--resample-seed") a.resample_seed = strtoull(val().c_str(), nullptr, 10);
        else if (f == "--no-kde-thinning") a.kde_thinning = !a.kde_thinning;
        else if (f == "--gpus") a.gpus = atoi(val().c_str());
        else if (f == "--devices") {   // explicit HIP ordinals, e.g. 0,1,2,3 (an ordinal may repeat: shards share that GPU)
            std::stringstream ss(val());
            std::string tok;
            while (std::getline(ss, tok, ',')) a.devices.push_back(atoi(tok.c_str()));
        }
        else usage(("unknown flag " + f).c_str());
    }
    // validators of src/garlic-cli.cpp:240-462 that concern Phase I
    if (a.tped.empty() || a.tfam.empty()) usage("--tped and --tfam are required");
    if (a.build == "none" && a.centromere == "none") usage("must provide --build or --centromere (garlic-cli.cpp:285-291)");
    if ((a.error <= 0 || a.error >= 1) && a.tgls == "none")
        usage("Genotype error rate must be > 0 and < 1, or a TGLS file must be provided.");
    if (a.tgls != "none" && a.gl_type != "GQ" && a.gl_type != "GL" && a.gl_type != "PL")
        usage("Must choose GQ/GL/PL for genotype likelihood format.");
    if (a.winsize <= 1 && a.winsize_multi.empty()) usage("SNP window size must be > 1.");
    for (int w : a.winsize_multi) if (w <= 1) usage("SNP window sizes must be > 1.");
    if (a.max_gap < 0) usage("Max gap must be > 0.");
    if (a.overlap_frac < 0 || a.overlap_frac > 1) usage("Overlap fraction must be >= 0 and <= 1.");
    if (a.weighted && a.map == "none") usage("--weighted needs --map");
    if (a.cm && a.map == "none") usage("--cm needs --map");
    if (a.have_cutoff && a.size_bounds.empty())
        usage("--lod-cutoff writes the ROH calls and needs --size-bounds (the size classes otherwise come from GARLIC's GMM stage, "
              "Phase II, not part of this tool)");
    for (size_t k = 1; k < a.size_bounds.size(); k++)
        if (!(a.size_bounds[k] > a.size_bounds[k - 1])) usage("--size-bounds must increase");
    return a;
}

void writeFeed(const std::string &path, const DoubleData *d)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { std::cerr << "ERROR: cannot write " << path << "\n"; throw 0; }
    fwrite(d->data, sizeof(double), (size_t)d->size, f);
    fclose(f);
    std::cerr << "Wrote " << path << " (" << d->size << " window scores)\n";
}
} // namespace

int main(int argc, char **argv)
{
    const Args a = parse(argc, argv);
    try {
        centromere centro(a.build, a.centromere, "none");
        int numLoci = 0, numInd = 0;
        std::vector<HapData *> *haps = nullptr; std::vector<MapData *> *maps = nullptr;
        std::vector<FreqData *> *freqs = nullptr; std::vector<GenoLikeData *> *gls = nullptr;
        IndData *ind = nullptr;
        struct Owner {   // the reference-shaped containers are raw pointers: release them on every way out
            std::vector<HapData *> *&haps; std::vector<MapData *> *&maps; std::vector<FreqData *> *&freqs;
            std::vector<GenoLikeData *> *&gls; IndData *&ind;
            ~Owner()
            {
                if (ind) releaseIndData(ind);
                if (haps) releaseHapData(haps);
                if (maps) releaseMapData(maps);
                if (freqs) releaseFreqData(freqs);
                if (gls) releaseGLData(gls);
            }
        } owner{haps, maps, freqs, gls, ind};
        FILE *probe = a.cache == "none" ? nullptr : fopen(a.cache.c_str(), "rb");
        if (probe) {   // parsed before: load the 2-bit sidecar instead of the text
            fclose(probe);
            loadGenotypeCache(a.cache, numLoci, numInd, &haps, &maps, &freqs, /*keepPacked=*/true);
            std::cerr << "Loaded genotype cache " << a.cache << "\n";
            if (a.phased && !haps->at(0)->firstCopy) {
                std::cerr << "ERROR: --phased, but " << a.cache << " was written without phase; delete it to re-read the tped\n";
                return 1;
            }
        } else {
            loadTPEDData(a.tped, numLoci, numInd, &haps, &maps, &freqs, a.tped_missing, a.phased, a.resample,
                         a.resample_seed);
            if (a.resample > 0) std::cerr << "Allele frequencies resampled: " << a.resample << "\n";   // garlic-main.cpp:91
            if (a.cache != "none") {
                writeGenotypeCache(a.cache, haps, maps, freqs);
                std::cerr << "Wrote genotype cache " << a.cache << "\n";
            }
        }
        std::string pop;
        int nindFam = 0;
        scanIndData3(a.tfam, nindFam, pop);
        if (nindFam != numInd) { std::cerr << "ERROR: tfam lists " << nindFam << " individuals, tped has " << numInd << "\n"; return 1; }
        ind = readIndData3(a.tfam, numInd);
        std::cerr << "Loaded " << numLoci << " loci x " << numInd << " individuals (" << maps->size() << " chromosomes)\n";

        const bool USE_GL = a.tgls != "none";
        if (USE_GL) gls = readTGLSData(a.tgls, numLoci, numInd, maps, a.gl_type, /*compact=*/true); // rows in pre-filter TPED order
        if (a.freq_file != "none") { releaseFreqData(freqs); freqs = nullptr; freqs = readFreqData(a.freq_file, maps); }
        else writeFreqData(a.out + ".freq", freqs, maps);                        // garlic-main.cpp:245-253
        int kept;
        if (a.weighted) {   // garlic-main.cpp:233-239, 267-276
            std::vector<GenMapScaffold *> *scaffold = loadMapScaffold(a.map, &centro);
            if (scaffold->size() != maps->size()) {
                std::cerr << "ERROR: Scaffold genetic map does not have the same number of chromosomes as data.\n";
                releaseGenMapScaffold(scaffold);
                return 1;
            }
            kept = filterMonomorphicAndOOBSites(&maps, &haps, &freqs, &gls, scaffold, USE_GL);
            const int ni = interpolateGeneticmap(maps, scaffold);
            std::cerr << "Number of genetic map locations interpolated: " << ni << "\n";
            releaseGenMapScaffold(scaffold);
        } else {
            kept = filterMonomorphicSites(&maps, &haps, &freqs, &gls, USE_GL);
        }
        std::cerr << "Filtered monomorphic" << (a.weighted ? " or out of bounds" : "") << " sites: " << kept << " loci kept\n";

        std::vector<int> devices = a.devices;
        if (devices.empty())
            for (int d = 0; d < a.gpus; d++) devices.push_back(d);

        std::vector<int> sizes = a.winsize_multi.empty() ? std::vector<int>{a.winsize} : a.winsize_multi;
        LodEngine engine(haps, freqs, maps, gls, &centro, USE_GL, devices); // one upload, many window sizes
        const std::vector<int> ldsub = a.weighted ? drawLdSubsample(numInd, a.ld_subsample, a.ld_seed) : std::vector<int>();
        // selectLODCutoff (garlic-roh.cpp:674-675): the KDE sees --kde-subsample individuals (default 20,
        // garlic-cli.cpp:131), 0 = everyone.  The ROH stage scores everyone; --raw-lod still writes all rows.
        const std::vector<int> kdesub = drawKdeSubsample(numInd, a.kde_subsample, a.kde_seed);
        if (!kdesub.empty()) {
            std::cerr << "Individuals used for KDE:";
            for (int i : kdesub) std::cerr << " " << ind->indID[i];
            std::cerr << "\n";
        }
        if (!a.raw_lod && !a.weighted && !USE_GL && sizes.size() > 1) {
            // --winsize-multi, feeds only, unweighted: all sizes in one call (their kernels and downloads overlap)
            std::vector<int> steps;
            for (int W : sizes) steps.push_back(a.kde_thinning ? W : 1);
            std::vector<DoubleData *> feeds = engine.lodFeedMulti(sizes, a.error, a.max_gap, &steps, &kdesub);
            for (size_t i = 0; i < sizes.size(); i++) {
                writeFeed(a.out + "." + std::to_string(sizes[i]) + "SNPs.lod.f64", feeds[i]);
                releaseDoubleData(feeds[i]);
            }
            sizes.clear();
        }
        // --lod-cutoff: the ROH calls (garlic-main.cpp:346-420 with a user cutoff and user size classes): assembleROHWindows
        // on the device(s), straight from the genotypes -- no window scores, no per-SNP counts -- then the .roh.bed
        auto roh_calls = [&](int W, bool single) {
            if (!a.have_cutoff) return;
            ROHLength *len = nullptr;
            std::vector<ROHData *> *roh = engine.assembleROHWindows(ind, a.lod_cutoff, &len, W, a.error, a.max_gap, a.overlap_frac,
                                                                    a.cm, a.weighted, a.M, a.mu);
            std::cerr << "ROH segments: " << (long long)len->size << " (garlic-lod, MI355X; .roh.bed in the format of garlic 1.1.6a)\n";
            writeROHData((single ? a.out : a.out + "." + std::to_string(W) + "SNPs") + ".roh.bed", roh, maps, a.size_bounds, ind->pop,
                         "1.1.6a", a.cm);      // the track lines name the format's version (garlic VERSION); this tool says who it is on stderr
            releaseROHData(roh);
            releaseROHLength(len);
        };
        const bool single_size = a.winsize_multi.empty();
        if (a.have_cutoff && !a.weighted)
            for (int W : a.winsize_multi.empty() ? std::vector<int>{a.winsize} : a.winsize_multi) roh_calls(W, single_size);
        for (int W : sizes) {
            if (a.weighted) engine.ldWeights(W, ldsub, false, a.phased);   // garlic-main.cpp:346-357: LD weights per window size
            if (a.weighted) roh_calls(W, single_size);
            const std::string feed_path = a.out + "." + std::to_string(W) + "SNPs.lod.f64";
            if (!a.raw_lod) {   // only the KDE feed is wanted: thin on the device, no full-score download
                DoubleData *feed = engine.lodFeed(W, a.error, a.max_gap, a.kde_thinning ? W : 1, a.weighted, a.M, a.mu, &kdesub);
                writeFeed(feed_path, feed);
                releaseDoubleData(feed);
                continue;
            }
            std::vector<WinData *> *win = a.weighted ? engine.wlodWindowsResident(W, a.error, a.max_gap, a.M, a.mu)
                                                     : engine.lodWindows(W, a.error, a.max_gap);
            writeWinData(win, ind, maps, sizes.size() == 1 ? a.out : a.out + "." + std::to_string(W) + "SNPs");
            DoubleData *feed = kdesub.empty() ? convertWinData2DoubleData(win, a.kde_thinning ? W : 1)
                                              : convertSubsetWinData2DoubleData(win, kdesub, a.kde_thinning ? W : 1);
            writeFeed(feed_path, feed);
            releaseDoubleData(feed);
            releaseWinData(win);
        }
        if (a.winsize_stream) {
            // selectWinsize (garlic-roh.cpp:766-850) computes a window size, looks at the KDE of its scores and, if that is
            // not smooth enough, goes on with winsize + --auto-winsize-step on the same data.  The KDE is Phase II and not
            // part of this tool, so its owner drives that loop: one window size per line on stdin ("+" = the previous size
            // + --auto-winsize-step), the feed of each on the resident panel, one line "FEED <winsize> <file> <values>"
            // on stdout when it is written; end of input or 0 ends the run.
            int last = a.winsize_multi.empty() ? a.winsize : a.winsize_multi.back();
            std::string line;
            while (std::getline(std::cin, line)) {
                while (!line.empty() && isspace((unsigned char)line.back())) line.pop_back();
                if (line.empty()) continue;
                const int W = line == "+" ? last + a.auto_winsize_step : atoi(line.c_str());
                if (W == 0) break;
                if (W <= 1) { std::cerr << "ERROR: SNP window size must be > 1.\n"; return 1; }
                if (a.weighted) engine.ldWeights(W, ldsub, false, a.phased);
                DoubleData *feed = engine.lodFeed(W, a.error, a.max_gap, a.kde_thinning ? W : 1, a.weighted, a.M, a.mu, &kdesub);
                const std::string path = a.out + "." + std::to_string(W) + "SNPs.lod.f64";
                writeFeed(path, feed);
                std::cout << "FEED " << W << " " << path << " " << feed->size << std::endl;
                releaseDoubleData(feed);
                last = W;
            }
        } else if (a.auto_winsize)
            std::cerr << "NOTE: --auto-winsize picks among the feeds above in GARLIC's KDE stage (Phase II, not part of this tool); "
                         "--winsize-stream lets that stage ask for further window sizes on the resident panel\n";
    } catch (...) {
        return 1;
    }
    return 0;
}
