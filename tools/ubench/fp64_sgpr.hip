// Does a 64-bit SGPR source slow an FP64 VALU instruction on gfx950?  One workgroup on one CU (no power cap in play),
// 1 .. 6 waves per SIMD, each wave a loop of 32 x { v_mul_f64 t, sc, W ; v_add_f64 a_r, a_r, t } with W an SGPR pair
// (mode 0), a VGPR pair (mode 1), or the pair a, b of blocks sharing one product register as in
// GARLIC_WLOD_GLF_LOOP_ASM (mode 2: mul, add, mul, add on ONE temporary, SGPR weights).  s_memtime ticks (100 MHz on this part) and
// wall clock cycles per FP64 instruction of a SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ void __launch_bounds__(1024) k(double *out, unsigned long long *ticks, int iters, double w0)
{
    double a[16], b[16], t, t2, sc = 1.0 + threadIdx.x * 1e-9, scb = 1.5 + threadIdx.x * 1e-9;
#pragma unroll
    for (int r = 0; r < 16; r++) { a[r] = threadIdx.x * 1e-3 + r; b[r] = a[r] + 1; }
    double wv = w0 + threadIdx.x * 1e-12;
    const double ws = w0;     // kernel argument: SGPR pair
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (MODE == 0)
                asm volatile("v_mul_f64 %0, %4, %6\n\tv_mul_f64 %1, %5, %6\n\tv_add_f64 %2, %2, %0\n\tv_add_f64 %3, %3, %1"
                             : "=&v"(t), "=&v"(t2), "+v"(a[r]), "+v"(b[r]) : "v"(sc), "v"(scb), "s"(ws));
            else if (MODE == 1)
                asm volatile("v_mul_f64 %0, %4, %6\n\tv_mul_f64 %1, %5, %6\n\tv_add_f64 %2, %2, %0\n\tv_add_f64 %3, %3, %1"
                             : "=&v"(t), "=&v"(t2), "+v"(a[r]), "+v"(b[r]) : "v"(sc), "v"(scb), "v"(wv));
            else
                asm volatile("v_mul_f64 %0, %3, %5\n\tv_add_f64 %1, %1, %0\n\tv_mul_f64 %0, %4, %5\n\tv_add_f64 %2, %2, %0"
                             : "=&v"(t), "+v"(a[r]), "+v"(b[r]) : "v"(sc), "v"(scb), "s"(ws));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int r = 0; r < 16; r++) s += a[r] + b[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) ticks[threadIdx.x >> 6] = t1 - t0;
}

template <int MODE>
void run(const char *name, double *d, unsigned long long *dt)
{
    const int iters = 2000;
    for (int wps = 1; wps <= 4; wps++) {          // waves per SIMD (1024 threads = 16 waves max per workgroup)
        const int threads = 256 * wps;
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, d, dt, iters, 0.5);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, d, dt, iters, 0.5);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[16];
        hipMemcpy(h, dt, sizeof(h), hipMemcpyDeviceToHost);
        const double instr_per_simd = (double)iters * 64 * wps;          // FP64 instructions a SIMD issued
        printf("%-34s waves/SIMD %d: %8.3f ms = %.2f ns per FP64 instr of a SIMD; wave 0: %llu ticks (%.3f per instr)\n", name, wps, ms,
               ms * 1e6 / instr_per_simd, h[0], (double)h[0] / ((double)iters * 64));
    }
}

int main()
{
    double *d;
    unsigned long long *dt;
    hipMalloc(&d, sizeof(double) * 1024);
    hipMalloc(&dt, sizeof(unsigned long long) * 16);
    run<0>("SGPR weight, two temporaries", d, dt);
    run<1>("VGPR weight, two temporaries", d, dt);
    run<2>("SGPR weight, one temporary", d, dt);
    return 0;
}
