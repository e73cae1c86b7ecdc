// CPU check of the host adapter's TGLS reader (both forms) against the reference's readTGLSData
// (garlic-data.cpp:1509-1590 through oracle/_ref/libgarlic_ref.so), bit for bit.
#include "../../garlic_amd/host/garlic_host.hpp"

#include <cstring>
#include <dlfcn.h>
#include <iostream>

using namespace garlic_host;
typedef int (*ref_readTGLS_t)(const char *, int, int, const char *, double *);

int main(int argc, char **argv)
{
    if (argc != 6) { std::cerr << "usage: tgls_unit libgarlic_ref.so tgls gl_type nloci nind\n"; return 2; }
    void *lib = dlopen(argv[1], RTLD_NOW);
    if (!lib) { std::cerr << dlerror() << "\n"; return 2; }
    ref_readTGLS_t ref = (ref_readTGLS_t)dlsym(lib, "ref_readTGLS");
    if (!ref) { std::cerr << "ref_readTGLS missing\n"; return 2; }
    const std::string path = argv[2], type = argv[3];
    const int nloci = atoi(argv[4]), nind = atoi(argv[5]);
    try {
        std::vector<double> want((size_t)nloci * nind);
        if (ref(path.c_str(), nloci, nind, type.c_str(), want.data()) != 0) { std::cerr << "reference reader failed\n"; return 1; }
        MapData *m = initMapData(nloci);
        std::vector<MapData *> maps{m};
        for (int compact = 0; compact < 2; compact++) {
            std::vector<GenoLikeData *> *g = readTGLSData(path, nloci, nind, &maps, type, compact != 0);
            for (int l = 0; l < nloci; l++)
                for (int i = 0; i < nind; i++) {
                    const double x = likelihoodAt(g->at(0), l, i);
                    if (memcmp(&x, &want[(size_t)l * nind + i], sizeof x) != 0) {
                        std::cerr << "compact " << compact << " locus " << l << " ind " << i << ": " << x << " vs "
                                  << want[(size_t)l * nind + i] << "\n";
                        return 1;
                    }
                }
            releaseGLData(g);
        }
        releaseMapData(m);
    } catch (...) { std::cerr << "exception\n"; return 1; }
    std::cout << "tgls_unit ok\n";
    return 0;
}
