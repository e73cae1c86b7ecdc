// How fast gfx950 evaluates lod() (src/garlic-roh.cpp:355-386 with glibc's log10 restated: garlic_amd/csrc/tgls_math.hpp)
// when nothing comes from memory: arguments in registers, the log table in LDS, every CU full.  This is what "compute
// the TGLS term from the 1-byte code in the chain kernel instead of reading 8 B of term"  would
// have to pay per window and lane; the chain kernel today reads the term at ~5 TB/s: 1.6 ps per term chip-wide.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../../garlic_amd/csrc/tgls_math.hpp"
using namespace garlic;

__global__ void __launch_bounds__(256) lod_rate_kernel(const double *tab_g, double *sink, int reps, double f0, double e0)
{
    __shared__ double tab[256];
    tab[threadIdx.x] = tab_g[threadIdx.x];
    __syncthreads();
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    double f = f0 + 1e-7 * (t & 1023), e = e0 * (1 + (t & 63)), acc = 0.0;
    uint32_t g = t % 3u;
    for (int r = 0; r < reps; r++) {
#pragma unroll 4
        for (int k = 0; k < 4; k++) {
            acc += lod_term(g, f, e, tab);
            f += 1e-9;                      // (new arguments every time: nothing hoistable)
            g = g == 2u ? 0u : g + 1u;
        }
    }
    sink[t] = acc;
}

int main()
{
    static const double tab[256] = GLIBC_LOG_TAB;
    double *d_tab, *d_sink;
    const int blocks = 256 * 8, threads = 256, reps = 2000;
    hipMalloc(&d_tab, sizeof tab);
    hipMalloc(&d_sink, sizeof(double) * blocks * threads);
    hipMemcpy(d_tab, tab, sizeof tab, hipMemcpyHostToDevice);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int it = 0; it < 3; it++) {
        hipEventRecord(a);
        hipLaunchKernelGGL(lod_rate_kernel, dim3(blocks), dim3(threads), 0, 0, d_tab, d_sink, reps, 0.05, 1e-6);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        const double terms = (double)blocks * threads * reps * 4;
        printf("lod() terms: %.3e in %.3f ms = %.3e terms/s chip-wide = %.2f ps per term; 1.25e10 terms (10M SNPs x 1250 individuals): %.1f ms\n",
               terms, ms, terms / (ms * 1e-3), ms * 1e-3 / terms * 1e12, 1.25e10 / (terms / (ms * 1e-3)) * 1e3);
    }
    return 0;
}
