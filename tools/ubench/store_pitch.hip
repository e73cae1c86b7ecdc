// Does the row pitch of the score matrix decide how fast HBM takes the chain kernel's write pattern?
// 256 workgroups (one wave each, one per CU), each owning 64 consecutive rows and writing 256 contiguous
// bytes of every row per step with non-temporal 16-B stores (4 rows x 256 B per instruction), all rows
// advancing together -- POST's pattern.  Sweeps the pitch in 256-B steps, on a physically contiguous
// allocation (deterministic address -> channel mapping) and on a plain one.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

typedef double f64x2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(64) k(double *out, int64_t pitch /*doubles*/, int iters)
{
    const int lane = threadIdx.x;
    const int rsub = lane >> 4, csub = lane & 15;
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    f64x2 v = {(double)lane, (double)blockIdx.x};
    char *base = reinterpret_cast<char *>(out + row0 * pitch);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 16; q++) {
            char *p = base + ((int64_t)(q * 4 + rsub) * pitch) * 8 + (int64_t)it * 256 + csub * 16;
            __builtin_nontemporal_store(v, reinterpret_cast<f64x2 *>(p));
        }
    }
}

static double run(double *d, int64_t pitch, int nblocks, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(nblocks), dim3(64), 0, 0, d, pitch, iters);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(nblocks), dim3(64), 0, 0, d, pitch, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return (double)nblocks * 64 * iters * 256 / best / 1e9;   // TB/s
}

int main(int argc, char **argv)
{
    const int64_t p0 = argc > 1 ? atoll(argv[1]) : 86528;     // doubles: ~ chromosome 1 of the C2 panel
    const int nblocks = 256, iters = 2048;                     // 512 KB of every row
    const int64_t maxpitch = p0 + 32 * 160;
    const size_t bytes = (size_t)nblocks * 64 * maxpitch * 8;
    for (int contiguous = 1; contiguous >= 0; contiguous--) {
        double *d = nullptr;
        hipError_t e = contiguous ? hipExtMallocWithFlags((void **)&d, bytes, hipDeviceMallocContiguous) : hipMalloc(&d, bytes);
        if (e != hipSuccess) { printf("alloc (contiguous=%d) failed: %s\n", contiguous, hipGetErrorString(e)); continue; }
        printf("contiguous=%d base %p\n", contiguous, (void *)d);
        for (int kpad = 0; kpad <= 160; kpad += (kpad < 40 ? 1 : 8)) {
            const int64_t pitch = p0 + 32 * kpad;
            printf("  pitch %8lld doubles (%9lld B, /256 = %6lld, mod 4K=%4lld mod 32K=%5lld mod 1M=%7lld): %.2f TB/s\n",
                   (long long)pitch, (long long)pitch * 8, (long long)pitch * 8 / 256, (long long)(pitch * 8 % 4096),
                   (long long)(pitch * 8 % 32768), (long long)(pitch * 8 % (1 << 20)), run(d, pitch, nblocks, iters));
        }
        hipFree(d);
    }
    return 0;
}
