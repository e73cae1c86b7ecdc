// CPU check of the host adapter's genetic-map path (loadMapScaffold + interpolateGeneticmap) against the
// real reference functions in oracle/_ref/libgarlic_ref.so (ref_interpolate), bit for bit.
#include "../../garlic_amd/host/garlic_host.hpp"

#include <cstring>
#include <dlfcn.h>
#include <iostream>

using namespace garlic_host;
typedef int (*ref_interpolate_t)(const char *, int, int, int, int, const int *, double *);

int main(int argc, char **argv)
{
    if (argc < 5) { std::cerr << "usage: map_unit libgarlic_ref.so mapfile chr pos...\n"; return 2; }
    void *lib = dlopen(argv[1], RTLD_NOW);
    if (!lib) { std::cerr << dlerror() << "\n"; return 2; }
    ref_interpolate_t ref = (ref_interpolate_t)dlsym(lib, "ref_interpolate");
    if (!ref) { std::cerr << "ref_interpolate missing\n"; return 2; }
    const std::string mapfile = argv[2], chr = argv[3];
    std::vector<int> pos;
    for (int i = 4; i < argc; i++) pos.push_back(atoi(argv[i]));
    try {
        centromere centro("none", "none", "none");
        centro.set(chr, 0, 0);
        std::vector<GenMapScaffold *> *sc = loadMapScaffold(mapfile, &centro);
        MapData *m = initMapData((int)pos.size());
        m->chr = sc->at(0)->chr;
        for (size_t l = 0; l < pos.size(); l++) m->physicalPos[l] = pos[l];
        std::vector<MapData *> maps{m};
        const int mine = interpolateGeneticmap(&maps, sc);
        std::vector<double> want(pos.size());
        const int theirs = ref(mapfile.c_str(), 0, 0, 1, (int)pos.size(), pos.data(), want.data());
        if (mine != theirs) { std::cerr << "interpolated " << mine << " vs " << theirs << "\n"; return 1; }
        for (size_t l = 0; l < pos.size(); l++)
            if (memcmp(&want[l], &m->geneticPos[l], sizeof(double)) != 0) {
                std::cerr << "site " << l << " pos " << pos[l] << ": " << m->geneticPos[l] << " vs " << want[l] << "\n";
                return 1;
            }
        releaseGenMapScaffold(sc);
        releaseMapData(m);
    } catch (...) { std::cerr << "exception\n"; return 1; }
    std::cout << "map_unit ok\n";
    return 0;
}
