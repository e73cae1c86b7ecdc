// CPU check of garlic_amd/csrc/tgls_math.hpp: the restatement of glibc's log10 (and lod() on top of it)
// that the TGLS kernels run on the device, compiled here for the host and compared bit for bit with the
// host libm's log10 -- the function the reference calls (src/garlic-roh.cpp:385).  Every operation in
// that header is a correctly rounded IEEE operation on both machines, so agreement here is agreement
// of the algorithm; the -m gpu tests and the library's start-up check cover the device's arithmetic.
//   usage: log10_unit [n_random]      (exit 0 and "log10_unit ok" when everything agrees)
#include "../../garlic_amd/csrc/tgls_math.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

using namespace garlic;

static const double TAB[256] = GLIBC_LOG_TAB;
static long long n_checked = 0, n_bad = 0;

static void check(double x)
{
    const double want = log10(x), got = glibc_log10(x, TAB);
    n_checked++;
    if (f64_bits(want) != f64_bits(got)) {
        if (n_bad < 20)
            fprintf(stderr, "log10(%a): libm %a (%016llx), port %a (%016llx)\n", x, want,
                    (unsigned long long)f64_bits(want), got, (unsigned long long)f64_bits(got));
        n_bad++;
    }
}

// lod() as the reference writes it (src/garlic-roh.cpp:355-386), host libm
static double ref_lod(int genotype, double freq, double error)
{
    double autozygous = 1, nonAutozygous = 1;
    if (freq == 0 || freq == 1) {
    } else if (genotype == 0) {
        nonAutozygous = (1 - freq) * (1 - freq);
        autozygous = (1 - error) * (1 - freq) + error * nonAutozygous;
    } else if (genotype == 1) {
        nonAutozygous = 2 * (freq) * (1 - freq);
        autozygous = error * nonAutozygous;
    } else if (genotype == 2) {
        nonAutozygous = (freq) * (freq);
        autozygous = (1 - error) * (freq) + error * nonAutozygous;
    }
    return log10(autozygous / nonAutozygous);
}

static void check_lod(int g, double f, double e)
{
    const double want = ref_lod(g, f, e), got = lod_term(g < 0 || g > 2 ? 3u : (uint32_t)g, f, e, TAB);
    n_checked++;
    if (f64_bits(want) != f64_bits(got)) {
        if (n_bad < 20)
            fprintf(stderr, "lod(%d, %a, %a): libm %a (%016llx), port %a (%016llx)\n", g, f, e, want,
                    (unsigned long long)f64_bits(want), got, (unsigned long long)f64_bits(got));
        n_bad++;
    }
}

int main(int argc, char **argv)
{
    const long long n_random = argc > 1 ? atoll(argv[1]) : 4000000;
    std::mt19937_64 rng(20260105);
    // specials
    const double inf = INFINITY;
    for (double x : {0.0, -0.0, 1.0, -1.0, inf, -inf, (double)NAN, -(double)NAN, 5e-324, 2.2250738585072014e-308,
                     2.225073858507201e-308, 1.7976931348623157e308, 0.5, 2.0, 10.0, 100.0, 1e-16, 0.9375, 1.064697265625})
        check(x);
    check(f64_from_bits(0x7FF0000000000001ull));   // signalling NaNs come back quieted
    check(f64_from_bits(0xFFF4000000000123ull));
    // every boundary of the algorithm +- a few ulps: the near-1 interval, the table cells around OFF
    for (uint64_t c : {0x3FEE000000000000ull, 0x3FF1090000000000ull, 0x3FF0000000000000ull, 0x3FE6000000000000ull,
                       0x3FF6000000000000ull, 0x3FE0000000000000ull, 0x0010000000000000ull})
        for (int d = -40; d <= 40; d++) check(f64_from_bits(c + (uint64_t)(int64_t)d));
    for (int cell = 0; cell < 128; cell++)          // both ends of each of the 128 table cells, 2 binades
        for (uint64_t top : {0x3FE0000000000000ull, 0x3FF0000000000000ull})
            for (int d = -8; d <= 8; d++) check(f64_from_bits(top + ((uint64_t)cell << 45) + (uint64_t)(int64_t)d));
    // random mantissas in the two binades log10 hands to log, dense around 1, then random exponents
    for (long long n = 0; n < n_random; n++) {
        const uint64_t m = rng() & 0x000FFFFFFFFFFFFFull;
        check(f64_from_bits(0x3FE0000000000000ull | m));
        check(f64_from_bits(0x3FF0000000000000ull | m));
        check(f64_from_bits((0x3FF0000000000000ull - (1ull << 49)) + (rng() % (3ull << 49))));   // [0.875, 1.25)
        check(f64_from_bits(((rng() % 2046 + 1) << 52) | m));                                     // any normal
        if ((n & 1023) == 0) check(f64_from_bits(m));                                             // subnormal
    }
    // lod(): the README's GQ / GL / PL examples through readTGLSData's conversion, grids, random
    const double readme[] = {pow(10, 30 / -10.0), 1 - pow(10, -0.000434511774018), 1 - pow(10, 0.00434511774018 / -10.0),
                             1e-16, 1.0, 0.001, 0.5};
    for (double e : readme)
        for (int g : {0, 1, 2, -9, 3})
            for (double f : {0.0, 1.0, 1e-6, 0.001, 0.25, 0.5, 0.75, 0.999, -0.25, 1.5, (double)NAN}) check_lod(g, f, e);
    for (int g : {0, 1, 2}) {
        check_lod(g, 0.3, NAN);
        check_lod(g, 0.3, -(double)NAN);
        check_lod(g, -(double)NAN, 0.01);
        check_lod(g, 0.3, inf);          // inf - inf inside: the default NaN
        check_lod(g, 0.3, -inf);
        check_lod(g, 0.3, 0.0);
        check_lod(g, 0.3, -0.5);
        check_lod(g, 0.3, 7.0);
        check_lod(g, 1e-200, 0.01);      // f * f underflows to 0: x / 0
        check_lod(g, 1.0 - 1e-16, 0.01);
    }
    std::uniform_real_distribution<double> U(0.0, 1.0);
    for (long long n = 0; n < n_random; n++) {
        const double f = 0.001 + 0.998 * U(rng);
        double e = U(rng);
        switch (n & 3) {
        case 0: e = pow(10, -10.0 * e); break;             // log-uniform 1e-10 .. 1 (GQ-like)
        case 1: e = 1 - pow(10, -0.3 * e); break;          // GL-like
        case 2: e = 1 - pow(10, -6.0 * e); break;          // PL-like
        default: break;
        }
        check_lod((int)(rng() % 3), f, e);
    }
    printf("%lld comparisons, %lld mismatches\n", n_checked, n_bad);
    if (n_bad) return 1;
    printf("log10_unit ok\n");
    return 0;
}
