// log, sin and cos for the gaussians of `velocity create` (nm_kernels.h: gaussian_fill).  The device library's versions are written for
// every argument a double can hold — Payne-Hanek reduction, double-double intermediates: ~850 instructions for one Box-Muller pair,
// a quarter of them v_add_f64 — and the draw is made once per atom and Hamiltonian move.  The arguments here are narrow: log of
// 1 - u in [2^-53, 1], sin and cos of t = 2 pi u in [0, 2 pi).  These are the classical fdlibm evaluations for such ranges (frexp,
// s = f / (2 + f) and a degree-14 polynomial; two-term Cody-Waite reduction by pi / 2 and the kernel polynomials on [-pi/4, pi/4]),
// with fused multiply-adds: within 1 ulp of the correctly rounded result (scripts/nm_math_check.cpp compares with libm over 2e7
// random arguments), as the device library's are.  Plain C++: the same text compiles for the host check.
#pragma once
#include <cmath>
#ifndef NM_HD
#define NM_HD __device__ __forceinline__
#endif
namespace nm {

// log(x) for 2^-1022 <= x <= 1 (normal, positive)
NM_HD double log_pos(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01, Lg7 = 1.479819860511658591e-01;
    int k;
    double m = frexp(x, &k);                       // x = m 2^k, 1/2 <= m < 1
    if (m < 0.70710678118654752440) { m *= 2.0; k -= 1; } // sqrt(1/2) <= m < sqrt(2)
    const double f = m - 1.0, dk = (double)k;
    const double s = f / (2.0 + f), z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, Lg6, Lg4), Lg2);
    const double t2 = z * fma(w, fma(w, fma(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1, hfsq = 0.5 * f * f;
    return dk * ln2_hi - ((hfsq - fma(s, hfsq + R, dk * ln2_lo)) - f);
}

// sin(t) and cos(t) for 0 <= t < 2 pi
NM_HD void sincos_2pi(double t, double &sn, double &cs)
{
    const double two_over_pi = 6.36619772367581382433e-01, pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
    const double kf = rint(t * two_over_pi);        // 0 .. 4
    double r = fma(-kf, pio2_hi, t);
    r = fma(-kf, pio2_lo, r);                       // |r| <= pi/4 (+ an ulp)
    const double z = r * r;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double ps = fma(z, fma(z, fma(z, fma(z, S6, S5), S4), S3), S2);
    const double s = fma(z * r, fma(z, ps, S1), r);
    const double pc = z * fma(z, fma(z, fma(z, fma(z, fma(z, C6, C5), C4), C3), C2), C1);
    const double hz = 0.5 * z, w = 1.0 - hz;
    const double c = w + (((1.0 - w) - hz) + z * pc);
    const int n = (int)kf & 3;
    sn = (n & 1) ? c : s; cs = (n & 1) ? s : c;
    if (n & 2) sn = -sn;
    if ((n + 1) & 2) cs = -cs;
}

} // namespace nm
