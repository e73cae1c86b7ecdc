// Practical FP64 VALU ceiling for the wLOD inner step: per lane 16 independent
// {v_mul_f64 (SGPR weight), v_add_f64} pairs per iteration, no memory traffic.  Reports pairs/s
// (one pair = one window term) for several waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(256) k(double *out, int iters, double w0)
{
    double acc[16];
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = threadIdx.x * 1e-3 + r;
    double sc = 1.0 + threadIdx.x * 1e-9;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] += sc * (w0 + r);   // w0 + r: wave-uniform -> SGPR
        sc += 1e-12;
    }
    double s = 0;
#pragma unroll
    for (int r = 0; r < 16; r++) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    double *d;
    hipMalloc(&d, sizeof(double) * 256 * 256 * 8 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps = 1; wps <= 8; wps *= 2) {          // waves per SIMD
        const int blocks = 256 * wps;                  // 256 CUs x (wps blocks of 4 waves)
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 0.5);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 0.5);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double pairs = (double)blocks * 256 * iters * 16;
        printf("waves/SIMD %d: %.3f ms  %.3e pairs/s  (%.1f TFLOP/s as mul+add)\n", wps, ms,
               pairs / ms * 1e3, 2 * pairs / ms * 1e3 / 1e12);
    }
    return 0;
}
