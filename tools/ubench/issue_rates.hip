// Micro-benchmarks of single-wave issue/latency on gfx950 (numbers feed tools/gen_chain_asm.py's schedule).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define REP256(x) REP4(REP64(x))

template <int MODE>
__global__ void __launch_bounds__(64) k(uint64_t *out, double *sink, double a, double b)
{
    __shared__ double lds[4096];
    lds[threadIdx.x] = a; lds[threadIdx.x + 64] = b;
    __syncthreads();
    double x = a, y = b, z = a + 1, w = b + 1;
    uint32_t i0 = threadIdx.x * 8, i1 = 7;
    uint64_t t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (MODE == 0) { asm volatile(REP256("v_add_f64 %0, %0, %1\n\t") : "+v"(x) : "v"(y)); }
    if (MODE == 1) { asm volatile(REP64("v_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %4\n\tv_add_f64 %2, %2, %4\n\tv_add_f64 %3, %3, %4\n\t") : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(a)); }
    if (MODE == 2) { asm volatile(REP256("v_add_f64 %0, %0, %1\n\tv_and_b32 %2, 24, %2\n\tv_and_b32 %3, 24, %3\n\t") : "+v"(x), "+v"(y), "+v"(i0), "+v"(i1)); }
    if (MODE == 3) { asm volatile(REP256("v_and_b32 %0, 24, %0\n\t") : "+v"(i0)); }
    if (MODE == 4) { asm volatile(REP256("v_and_b32 %0, 24, %1\n\t") : "+v"(i0) : "v"(i1)); }
    if (MODE == 5) { asm volatile(REP256("s_nop 0\n\t")); }
    if (MODE == 6) { asm volatile(REP256("ds_read_b64 %0, %1\n\t") "s_waitcnt lgkmcnt(0)" : "=&v"(x) : "v"(i0) : "memory"); }
    if (MODE == 7) { asm volatile(REP64("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\t") : "=&v"(x) : "v"(i0) : "memory"); }
    if (MODE == 8) { asm volatile(REP256("v_add_f64 %0, %0, %1\n\tv_and_b32 %2, 24, %2\n\tv_and_b32 %3, 24, %3\n\tv_and_b32 %2, 24, %2\n\tv_and_b32 %3, 24, %3\n\t") : "+v"(x), "+v"(y), "+v"(i0), "+v"(i1)); }
    if (MODE == 9) { asm volatile(REP256("v_add_f64 %0, %0, %1\n\tv_and_b32 %2, 24, %2\n\t") : "+v"(x), "+v"(y), "+v"(i0), "+v"(i1)); }
    if (MODE == 10) { asm volatile(REP256("v_add_f64 %0, %0, %1\n\ts_nop 0\n\t") : "+v"(x), "+v"(y)); }
    if (MODE == 11) { asm volatile(REP256("v_add_f64 %0, %0, %1\n\tds_read_b64 %2, %3\n\t") "s_waitcnt lgkmcnt(0)" : "+v"(x), "+v"(y), "=&v"(z) : "v"(i0) : "memory"); }
    if (MODE == 12) { asm volatile("v_mov_b64 v[100:101], %0\n\tv_mov_b64 v[102:103], %0\n\t" REP64("ds_write_b128 %1, v[100:103]\n\t") "s_waitcnt lgkmcnt(0)" :: "v"(x), "v"(i0 * 2) : "memory", "v100", "v101", "v102", "v103"); }
    if (MODE == 13) { asm volatile(REP256("v_lshrrev_b32 %0, 3, %1\n\tv_and_b32 %0, 24, %0\n\t") : "+v"(i0) : "v"(i1)); }
    if (MODE == 20) { asm volatile(REP64("ds_read_b128 v[100:103], %0\n\t") "s_waitcnt lgkmcnt(0)" :: "v"(i0 * 2) : "memory", "v100", "v101", "v102", "v103"); }
    if (MODE == 21) { asm volatile("v_mov_b64 v[100:101], %0\n\t" REP64("ds_write_b64 %1, v[100:101]\n\t") "s_waitcnt lgkmcnt(0)" :: "v"(x), "v"(i0) : "memory", "v100", "v101"); }
    if (MODE == 22) { asm volatile("v_mov_b64 v[100:101], %0\n\tv_mov_b64 v[102:103], %0\n\t" REP64("ds_write_b128 %1, v[100:103]\n\tv_and_b32 %2, 24, %2\n\tv_and_b32 %2, 24, %2\n\tv_and_b32 %2, 24, %2\n\tv_and_b32 %2, 24, %2\n\tv_and_b32 %2, 24, %2\n\tv_and_b32 %2, 24, %2\n\t") "s_waitcnt lgkmcnt(0)" :: "v"(x), "v"(i0 * 2), "v"(i1) : "memory", "v100", "v101", "v102", "v103"); }
    if (MODE == 23) { asm volatile(REP64("ds_read_b64 v[100:101], %0\n\tds_read_b64 v[102:103], %0\n\tv_and_b32 %1, 24, %1\n\tv_and_b32 %1, 24, %1\n\tv_and_b32 %1, 24, %1\n\tv_and_b32 %1, 24, %1\n\tv_add_f64 %2, %2, %3\n\tv_add_f64 %2, %2, %3\n\t") "s_waitcnt lgkmcnt(0)" :: "v"(i0), "v"(i1), "v"(x), "v"(y) : "memory", "v100", "v101", "v102", "v103"); }
    if (MODE == 24) { asm volatile(REP16("ds_read_b64 v[100:101], %0\n\tds_read_b64 v[102:103], %0\n\tds_read_b64 v[100:101], %0\n\tds_read_b64 v[102:103], %0\n\tds_read_b64 v[100:101], %0\n\tds_read_b64 v[102:103], %0\n\tds_read_b64 v[100:101], %0\n\tds_read_b64 v[102:103], %0\n\t" REP16("v_and_b32 %1, 24, %1\n\t") REP4("v_add_f64 %2, %2, %3\n\tv_add_f64 %2, %2, %3\n\t")) "s_waitcnt lgkmcnt(0)" :: "v"(i0), "v"(i1), "v"(x), "v"(y) : "memory", "v100", "v101", "v102", "v103"); }
    if (MODE == 25) { asm volatile("v_mov_b64 v[100:101], %0\n\tv_mov_b64 v[102:103], %0\n\t" REP64("global_store_dwordx4 %1, v[100:103], %2\n\t") "s_waitcnt vmcnt(0)" :: "v"(x), "v"(i0 * 2), "s"(sink) : "memory", "v100", "v101", "v102", "v103"); }
    if (MODE == 26) { asm volatile(REP64("ds_read2_b64 v[100:103], %0 offset1:1\n\t") "s_waitcnt lgkmcnt(0)" :: "v"(i0 * 2) : "memory", "v100", "v101", "v102", "v103"); }
    if (MODE == 27) { asm volatile("v_mov_b64 v[100:101], %0\n\tv_mov_b64 v[102:103], %0\n\t" REP64("ds_write2_b64 %1, v[100:101], v[102:103] offset1:1\n\t") "s_waitcnt lgkmcnt(0)" :: "v"(x), "v"(i0 * 2) : "memory", "v100", "v101", "v102", "v103"); }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    sink[threadIdx.x] = x + y + z + w + i0 + i1;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char *name, int n_instr)
{
    uint64_t *d; double *s;
    hipMalloc(&d, 8 * 64); hipMalloc(&s, 8 * 64 * 64);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, d, s, 1.0, 1e-9);
    uint64_t h;
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-44s %6llu cycles  %.2f per instr (%d)\n", name, (unsigned long long)h, (double)h / n_instr, n_instr);
    hipFree(d); hipFree(s);
}

int main()
{
    run<5>("s_nop 0 x256", 256);
    run<3>("v_and dependent x256", 256);
    run<4>("v_and independent x256", 256);
    run<13>("v_lshrrev+v_and x256 pairs", 512);
    run<0>("v_add_f64 dependent x256", 256);
    run<1>("v_add_f64 4 chains x256", 256);
    run<10>("v_add_f64 dep + s_nop x256", 512);
    run<9>("v_add_f64 dep + 1 v_and x256", 512);
    run<2>("v_add_f64 dep + 2 v_and x256", 768);
    run<8>("v_add_f64 dep + 4 v_and x256", 1280);
    run<6>("ds_read_b64 back-to-back x256", 256);
    run<7>("ds_read_b64 + wait x64 (latency)", 64);
    run<11>("v_add_f64 dep + ds_read_b64 x256", 512);
    run<12>("ds_write_b128 x64", 64);
    run<20>("ds_read_b128 b2b x64", 64);
    run<26>("ds_read2_b64 b2b x64", 64);
    run<21>("ds_write_b64 b2b x64", 64);
    run<27>("ds_write2_b64 b2b x64", 64);
    run<22>("ds_write_b128 + 6 v_and x64", 448);
    run<23>("[2 ds_read_b64, 4 v_and, 2 dp] x64", 512);
    run<24>("[8 ds_read, 16 v_and, 8 dp] x16", 512);
    run<25>("global_store_dwordx4 b2b x64 (sink 1KB)", 64);
    return 0;
}
