// lod() with an arbitrary per-genotype error on the device: the TGLS path for continuous likelihoods
// (--gl-type GL / PL, reference src/garlic-data.cpp:1555-1577 -> src/garlic-roh.cpp:68,91-95,117,245).
//
// "Bit-identical to the reference" means log10 must be the host libm's log10, which is not correctly
// rounded.  glibc 2.35 computes log10(x) as  y*log10_2lo + ivln10*log(m) + y*log10_2hi  (e_log10.c;
// plain multiplies and adds) on top of its table-driven log (e_log.c, N = 128), and on x86-64 hosts
// with FMA the IFUNC resolver selects the variant compiled with -mfma, whose contractions are fixed in
// the shipped binary.  glibc_log() below restates that variant operation by operation (the order was
// read off the disassembly of __log_fma; every fma() here is one vfmadd there, every other operation
// one rounded IEEE operation), so the same inputs give the same bits on any IEEE machine -- including
// gfx950, whose v_fma_f64 / v_mul_f64 / v_add_f64 / division are correctly rounded and keep FP64
// denormals.  The same source compiles for the host (tests/host_unit/log10_unit.cpp runs it against
// the host's log10); the library also checks the device against the host at run time
// (garlic_hip.hip: device_log10_matches_host) and computes the terms on the host if they ever differ.
#pragma once
#include <stdint.h>
#include <string.h>

#include "glibc_log_data.inc"

#if defined(__HIPCC__)
#define GARLIC_HD __host__ __device__ __forceinline__
#else
#define GARLIC_HD inline
#endif

namespace garlic {

struct GlibcLogTab {        // what a kernel stages in LDS: tab[i] = {invc, logc}
    double t[256];
};

GARLIC_HD double f64_from_bits(uint64_t u)
{
    double d;
    memcpy(&d, &u, sizeof d);
    return d;
}
GARLIC_HD uint64_t f64_bits(double d)
{
    uint64_t u;
    memcpy(&u, &d, sizeof u);
    return u;
}

// x86's default NaN (what 0/0, inf-inf, (x-x)/(x-x) produce there): sign bit set.  gfx950 produces
// +NaN for the same operations, so invalid operations are canonicalised to this wherever the
// reference's arithmetic would have produced the x86 one.
constexpr uint64_t X86_DEFAULT_NAN = 0xFFF8000000000000ull;

// __log_fma for finite positive normal x in [0.5, 2) -- all log10 passes it.  tab: 128 x {invc, logc}.
GARLIC_HD double glibc_log_core(double x, const double *tab)
{
    const double A[5] = GLIBC_LOG_POLY;
    const double B[11] = GLIBC_LOG_POLY1;
    const uint64_t ix = f64_bits(x);
    // 1 - 2^-4 <= x < 1 + 0x1.09p-4: polynomial around 1
    if (ix - 0x3FEE000000000000ull < 0x0003090000000000ull) {   // LO = asuint64(1.0 - 0x1p-4); HI - LO, HI = asuint64(1.0 + 0x1.09p-4)
        if (ix == 0x3FF0000000000000ull) return 0.0;
        const double r = x - 1.0;
        double p2 = __builtin_fma(B[2], r, B[1]);
        double p3 = __builtin_fma(B[5], r, B[4]);
        const double r2 = r * r;
        double p5 = __builtin_fma(B[8], r, B[7]);
        p2 = __builtin_fma(r2, B[3], p2);
        p3 = __builtin_fma(r2, B[6], p3);
        const double r3 = r * r2;
        double p1 = __builtin_fma(r2, B[9], p5);
        p1 = __builtin_fma(r3, B[10], p1);
        p1 = __builtin_fma(p1, r3, p3);
        p1 = __builtin_fma(p1, r3, p2);
        const double two27 = 134217728.0;
        const double t = __builtin_fma(r, two27, r);          // r + r * 2^27
        const double rhi = __builtin_fma(-two27, r, t);       // ... - r * 2^27
        const double rhi2 = rhi * rhi;
        const double rlo = r - rhi;
        const double hi = __builtin_fma(rhi2, B[0], r);
        const double d = r - hi;
        const double s = r + rhi;
        double lo = __builtin_fma(rhi2, B[0], d);
        const double q = B[0] * rlo;
        lo = __builtin_fma(q, s, lo);
        const double y = __builtin_fma(p1, r3, lo);
        return y + hi;
    }
    const uint64_t tmp = ix - 0x3FE6000000000000ull;            // OFF
    const int i = (int)((tmp >> 45) & 127u);
    const int k = (int)((int64_t)tmp >> 52);
    const uint64_t iz = ix - (tmp & 0xFFF0000000000000ull);
    const double invc = tab[2 * i], logc = tab[2 * i + 1];
    const double z = f64_from_bits(iz);
    const double r = __builtin_fma(z, invc, -1.0);
    const double kd = (double)k;
    const double w = __builtin_fma(kd, GLIBC_LOG_LN2HI, logc);
    const double p12 = __builtin_fma(A[2], r, A[1]);
    const double hi = r + w;
    const double r2 = r * r;
    double lo = w - hi;
    lo = lo + r;
    lo = __builtin_fma(kd, GLIBC_LOG_LN2LO, lo);
    const double r3 = r * r2;
    double p = __builtin_fma(r, A[4], A[3]);
    lo = __builtin_fma(r2, A[0], lo);
    p = __builtin_fma(p, r2, p12);
    const double y = __builtin_fma(r3, p, lo);
    return y + hi;
}

// log10 of glibc 2.35 as a program calls it: the wrapper (math/w_log10_compat.c) in front of
// __ieee754_log10 (sysdeps/ieee754/dbl-64/e_log10.c), every special case included
GARLIC_HD double glibc_log10(double x, const double *tab)
{
    uint64_t ix = f64_bits(x);
    int64_t k = -1023;
    if ((int64_t)ix < 0x0010000000000000ll) {                   // x < 2^-1022 (or negative)
        if ((ix << 1) == 0) return -__builtin_inf();            // log(+-0) = -two54 / fabs(x) = -inf
        // a negative number or -inf: the log10 wrapper every caller goes through (w_log10_compat.c ->
        // __kernel_standard case 19) returns the constant NAN, sign bit clear; a NaN operand skips the
        // wrapper's test and comes back from (x - x) / (x - x) quieted, sign and payload kept
        if ((int64_t)ix < 0) return f64_from_bits(x != x ? (ix | 0x0008000000000000ull) : 0x7FF8000000000000ull);
        x *= 18014398509481984.0;                               // 2^54: subnormal, scale up
        ix = f64_bits(x);
        k = -1077;
    }
    if (ix > 0x7FEFFFFFFFFFFFFFull)                            // x + x: +inf, or the NaN quieted
        return x != x ? f64_from_bits(ix | 0x0008000000000000ull) : x;
    k += (int64_t)(ix >> 52);
    const int64_t i = (int64_t)((uint64_t)k >> 63);
    const double y = (double)(k + i);
    const double m = f64_from_bits((ix & 0x000FFFFFFFFFFFFFull) | ((uint64_t)(0x3FF - i) << 52));
    const double z = y * GLIBC_LOG10_2LO + GLIBC_IVLN10 * glibc_log_core(m, tab);
    return z + y * GLIBC_LOG10_2HI;
}

// lod(), src/garlic-roh.cpp:355-386, genotype as the panel's 2-bit code (3 = anything but 0/1/2).
// Compile with -ffp-contract=off: the reference build has no FMA, each operation below rounds.
GARLIC_HD double lod_term(uint32_t code, double freq, double error, const double *tab)
{
    double aut = 1, non = 1;
    if (freq == 0 || freq == 1 || code > 2u) return 0.0;       // log10(1 / 1)
    // A NaN input reaches the result quieted, sign and payload kept, through every operation below on
    // x86 (one NaN operand: that operand); made explicit so that nothing depends on how gfx950
    // propagates payloads.  (Both NaN: which one survives depends on the reference build's register
    // allocation; the frequency is taken.)
    if (freq != freq) return f64_from_bits(f64_bits(freq) | 0x0008000000000000ull);
    if (error != error) return f64_from_bits(f64_bits(error) | 0x0008000000000000ull);
    if (code == 0u) {
        non = (1 - freq) * (1 - freq);
        aut = (1 - error) * (1 - freq) + error * non;
    } else if (code == 1u) {
        non = 2 * (freq) * (1 - freq);
        aut = error * non;
    } else if (code == 2u) {
        non = (freq) * (freq);
        aut = (1 - error) * (freq) + error * non;
    }
    double q = aut / non;
    // no input is a NaN here, so a NaN can only come from an invalid operation (inf - inf, 0 / 0, ..):
    // the default NaN, whose sign differs between x86 and gfx950
    if (q != q) q = f64_from_bits(X86_DEFAULT_NAN);
    return glibc_log10(q, tab);
}

} // namespace garlic
