// host check of the block kernel's own log / sin / cos (neuralmelting_amd/csrc/nm_math.h) against libm, ulp statistics
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <random>
#define NM_HD
#include "nm_math.h"
static double ulp_of(double x) { int e; frexp(x, &e); return ldexp(1.0, e - 53); }
int main(int argc, char **argv)
{
    const long N = argc > 1 ? atol(argv[1]) : 20000000;
    std::mt19937_64 g(1);
    double worst_log = 0, worst_s = 0, worst_c = 0; const double twopi = 6.283185307179586476925286766559;
    for (long n = 0; n < N; ++n) {
        const uint64_t r = g();
        const double u = (double)(r >> 11) * 0x1p-53;   // [0, 1)
        const double x = 1.0 - u;
        const double l = nm::log_pos(x), lr = std::log(x);
        if (lr != 0.0) worst_log = std::fmax(worst_log, std::fabs(l - lr) / ulp_of(lr));
        else if (l != 0.0) { printf("log(1) = %g\n", l); return 1; }
        const double t = twopi * u;
        double s, c; nm::sincos_2pi(t, s, c);
        const double sr = std::sin(t), cr = std::cos(t);
        worst_s = std::fmax(worst_s, std::fabs(s - sr) / ulp_of(std::fabs(sr) > 1e-300 ? sr : 1e-300));
        worst_c = std::fmax(worst_c, std::fabs(c - cr) / ulp_of(std::fabs(cr) > 1e-300 ? cr : 1e-300));
    }
    // the small end of log's argument
    for (int e = 1; e <= 53; ++e) { const double x = ldexp(1.0, -e); worst_log = std::fmax(worst_log, std::fabs(nm::log_pos(x) - std::log(x)) / ulp_of(std::log(x))); }
    printf("worst error in ulps of the libm result: log %.2f  sin %.2f  cos %.2f\n", worst_log, worst_s, worst_c);
    return (worst_log <= 1.0 && worst_s <= 1.0 && worst_c <= 1.0) ? 0 : 2;
}
