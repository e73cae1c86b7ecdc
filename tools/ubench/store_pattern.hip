// How fast can the chip absorb the LOD output pattern?  Each wave owns 64 rows (pitch apart) and
// writes SEG contiguous bytes of each row per iteration, rows advancing together (like the chain
// kernel's transpose-tile write-out).  Reports TB/s for several segment widths / waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int SEG>   // bytes per row per iteration: 128 (the wLOD write-out), 256 (the chain's), 512, 1024
__global__ void __launch_bounds__(64) k(double *out, int64_t pitch /*doubles*/, int iters, int waves_per_blockrow)
{
    const int lane = threadIdx.x;
    constexpr int LPR = SEG / 16;          // lanes per row
    constexpr int RPI = 64 / LPR;          // rows per instruction
    const int rsub = lane / LPR, csub = lane % LPR;
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    double2 v = make_double2(lane, blockIdx.x);
    char *base = reinterpret_cast<char *>(out + row0 * pitch);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 64 / RPI; q++) {
            char *p = base + ((int64_t)(q * RPI + rsub) * pitch) * 8 + (int64_t)it * SEG + csub * 16;
            *reinterpret_cast<double2 *>(p) = v;
        }
    }
}

template <int SEG> void run(double *d, int64_t pitch, int nblocks, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<SEG>, dim3(nblocks), dim3(64), 0, 0, d, pitch, iters, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<SEG>, dim3(nblocks), dim3(64), 0, 0, d, pitch, iters, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double bytes = (double)nblocks * 64 * iters * SEG;
    printf("SEG %4d  blocks %5d  iters %6d : %.3f ms  %.2f TB/s\n", SEG, nblocks, iters, ms, bytes / ms / 1e9);
}

int main()
{
    const int64_t pitch = 50016;                 // doubles per row (~ chr1 arm of the C2 panel)
    const int64_t rows = 64 * 4096;
    double *d;
    if (hipMalloc(&d, rows * pitch * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    for (int nb : {256, 512, 1024, 2048, 4096}) {
        run<128>(d, pitch, nb, (int)(pitch * 8 / 128));
        run<256>(d, pitch, nb, (int)(pitch * 8 / 256));
        run<512>(d, pitch, nb, (int)(pitch * 8 / 512));
        run<1024>(d, pitch, nb, (int)(pitch * 8 / 1024));
    }
    hipFree(d);
    return 0;
}
