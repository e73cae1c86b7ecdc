// What one chain wave's window costs on gfx950 (tools/gen_feed_asm.py): single wave, s_memtime around unrolled bodies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define REP256(x) REP4(REP64(x))
#define SD_LO(d, s, b) "v_lshlrev_b32_sdwa " d ", 4, " s " dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" b "\n\t"
#define SD_HI(d, s, b) "v_and_b32_sdwa " d ", s20, " s " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" b "\n\t"
// one window of the feed loop: offset of a later window, acc - t_out, that window's look-up, + t_in (buffers rotate over 8 quads v[40..71])
#define WIN_FEED(BN, BC, AP, AD) SD_LO("v" #BN, "v100", "1") "v_add_f64 v[" #AD ":" #AD "+1], v[" #AP ":" #AP "+1], -v[" #BC ":" #BC "+1]\n\t" \
    "ds_read_b128 v[" #BN ":" #BN "+3], v" #BN " offset:1024\n\t" "v_add_f64 v[" #AD ":" #AD "+1], v[" #AD ":" #AD "+1], v[" #BC "+2:" #BC "+3]\n\t"
#define WIN_BITS(BN, BC, AP, AD) SD_LO("v" #BN, "v100", "1") "v_add_f64 v[" #AD ":" #AD "+1], v[" #AP ":" #AP "+1], -v[" #BC ":" #BC "+1]\n\t" \
    "v_addc_co_u32_e32 v102, vcc, v102, v102, vcc\n\t" "v_add_f64 v[" #AD ":" #AD "+1], v[" #AD ":" #AD "+1], v[" #BC "+2:" #BC "+3]\n\t" \
    "ds_read_b128 v[" #BN ":" #BN "+3], v" #BN " offset:1024\n\t" "v_cmp_le_f64_e32 vcc, s[22:23], v[" #AD ":" #AD "+1]\n\t"
#define WIN_FEED64(BN, BC, AP, AD) SD_LO("v104", "v100", "1") "v_add_f64 v[" #AD ":" #AD "+1], v[" #AP ":" #AP "+1], -v[" #BC ":" #BC "+1]\n\t" \
    "ds_read_b64 v[" #BN ":" #BN "+1], v104 offset:1024\n\t" "ds_read_b64 v[" #BN "+2:" #BN "+3], v104 offset:1032\n\t" "v_add_f64 v[" #AD ":" #AD "+1], v[" #AD ":" #AD "+1], v[" #BC "+2:" #BC "+3]\n\t"
// 8 windows, look-up 6 windows ahead of its use, a counted wait every 4 windows
#define EIGHT(W) W(64, 40, 86, 72) W(68, 44, 72, 74) "s_waitcnt lgkmcnt(5)\n\t" W(40, 48, 74, 76) W(44, 52, 76, 78) W(48, 56, 78, 80) W(52, 60, 80, 82) "s_waitcnt lgkmcnt(5)\n\t" W(56, 64, 82, 84) W(60, 68, 84, 86)
#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63", \
    "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v100","v101","v102","v103","v104","v105", \
    "s20","s21","s22","s23","vcc","memory"

template <int MODE>
__global__ void __launch_bounds__(64) k(uint64_t *out, double *sink, double a, double b)
{
    __shared__ double lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = a * i;
    __syncthreads();
    uint64_t t0, t1;
    asm volatile("s_movk_i32 s20, 0xf0\n\ts_mov_b64 s[22:23], 0\n\tv_mov_b32 v100, 0x10203040\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v104, 0\n\t"
                 "v_mov_b32 v40, 0\n\tv_mov_b32 v44, 0\n\tv_mov_b32 v48, 0\n\tv_mov_b32 v52, 0\n\tv_mov_b32 v56, 0\n\tv_mov_b32 v60, 0\n\tv_mov_b32 v64, 0\n\tv_mov_b32 v68, 0\n\t"
                 "v_mov_b64 v[86:87], 0\n\tv_mov_b64 v[72:73], 0\n\t" ::: CLOB);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (MODE == 0) asm volatile(REP256(SD_LO("v101", "v100", "1")) ::: CLOB);
    if (MODE == 1) asm volatile(REP256(SD_HI("v101", "v100", "2")) ::: CLOB);
    if (MODE == 2) asm volatile(REP256("v_cmp_le_f64_e32 vcc, s[22:23], v[72:73]\n\tv_addc_co_u32_e32 v102, vcc, v102, v102, vcc\n\t") ::: CLOB);
    if (MODE == 3) asm volatile(REP64("ds_read_b128 v[40:43], v104 offset:1024\n\tv_and_b32 v101, 24, v100\n\tv_and_b32 v101, 24, v100\n\tv_and_b32 v101, 24, v100\n\t") "s_waitcnt lgkmcnt(0)" ::: CLOB);
    if (MODE == 4) asm volatile(REP64("ds_read_b128 v[40:43], v104 offset:1024\n\tv_add_f64 v[72:73], v[86:87], v[86:87]\n\tv_add_f64 v[74:75], v[86:87], v[86:87]\n\tv_add_f64 v[76:77], v[86:87], v[86:87]\n\tv_add_f64 v[78:79], v[86:87], v[86:87]\n\t") "s_waitcnt lgkmcnt(0)" ::: CLOB);
    if (MODE == 5) asm volatile(REP16(EIGHT(WIN_FEED)) "s_waitcnt lgkmcnt(0)" ::: CLOB);
    if (MODE == 6) asm volatile(REP16(EIGHT(WIN_BITS)) "s_waitcnt lgkmcnt(0)" ::: CLOB);
    if (MODE == 7) asm volatile(REP256("s_waitcnt lgkmcnt(15)\n\tv_and_b32 v101, 24, v100\n\t") ::: CLOB);
    if (MODE == 8) asm volatile(REP16(EIGHT(WIN_FEED64)) "s_waitcnt lgkmcnt(0)" ::: CLOB);
    if (MODE == 9) asm volatile(REP64("ds_read_b128 v[40:43], v104 offset:1024\n\tv_and_b32 v101, 24, v100\n\tv_and_b32 v101, 24, v100\n\tv_and_b32 v101, 24, v100\n\tv_and_b32 v101, 24, v100\n\tv_and_b32 v101, 24, v100\n\tv_and_b32 v101, 24, v100\n\tv_and_b32 v101, 24, v100\n\t") "s_waitcnt lgkmcnt(0)" ::: CLOB);
    if (MODE == 10) asm volatile(REP64("ds_read_b64 v[40:41], v104 offset:1024\n\tds_read_b64 v[42:43], v104 offset:1032\n\tv_and_b32 v101, 24, v100\n\tv_and_b32 v101, 24, v100\n\tv_and_b32 v101, 24, v100\n\t") "s_waitcnt lgkmcnt(0)" ::: CLOB);
    if (MODE == 11) asm volatile(REP256("s_barrier\n\tv_and_b32 v101, 24, v100\n\t") ::: CLOB);
    if (MODE == 12) asm volatile(REP64("s_cmp_lg_u32 s20, 7\n\ts_cbranch_scc0 1f\n\tv_and_b32 v101, 24, v100\n\t1:\n\t") ::: CLOB);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    sink[threadIdx.x] = lds[threadIdx.x];
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char *name, int n_units, const char *unit)
{
    uint64_t *d; double *s;
    hipMalloc(&d, 8 * 64); hipMalloc(&s, 8 * 64 * 64);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, d, s, 1.0, 1e-9);
    uint64_t h;
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    // s_memtime counts at 100 MHz on gfx950: convert with the shader clock the wave ran at? no: report ticks and ns
    printf("%-58s %7llu ticks  %.2f per %s (%d)\n", name, (unsigned long long)h, (double)h / n_units, unit, n_units);
    hipFree(d); hipFree(s);
}

int main()
{
    run<0>("v_lshlrev_b32_sdwa (byte -> byte 0) x256", 256, "instr");
    run<1>("v_and_b32_sdwa (sgpr, byte) x256", 256, "instr");
    run<2>("v_cmp_le_f64 -> vcc -> v_addc x256", 256, "pair");
    run<3>("[ds_read_b128 + 3 v_and] x64", 64, "group");
    run<9>("[ds_read_b128 + 7 v_and] x64", 64, "group");
    run<10>("[2 ds_read_b64 + 3 v_and] x64", 64, "group");
    run<4>("[ds_read_b128 + 4 independent v_add_f64] x64", 64, "group");
    run<5>("feed window body x128 (look-ahead 6)", 128, "window");
    run<8>("feed window body, 2 x ds_read_b64, x128", 128, "window");
    run<6>("bits window body x128 (look-ahead 6)", 128, "window");
    run<7>("[s_waitcnt lgkmcnt(15) + v_and] x256", 256, "pair");
    run<11>("[s_barrier (one wave) + v_and] x256", 256, "pair");
    run<12>("[s_cmp + s_cbranch (taken) ] x64", 64, "group");
    return 0;
}
