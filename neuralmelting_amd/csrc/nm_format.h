// nm_format.h — '%.4E' formatting of the reference's output files on the host, byte-identical to Python's / C's printf.
// A recorded cycle of C5 is 1024 x 2048 lines of text; Python's per-atom loop (remcmc:255-256) then dominates the run.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstring>

namespace nm {

// writes v like "%.4E" (no leading space); returns the length.  Five significant digits, correctly rounded (ties to even on
// the exact binary value): fast path by one correctly rounded division by an exact power of ten, snprintf whenever the
// fifth digit's rounding is not beyond doubt.
inline int fmt_e4(double v, char *o)
{
    static const double P10[23] = { 1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16,
                                    1e17, 1e18, 1e19, 1e20, 1e21, 1e22 };
    if (!(v == v)) { std::memcpy(o, "NAN", 3); return 3; }
    char *q = o;
    if (std::signbit(v)) { *q++ = '-'; v = -v; }
    if (std::isinf(v)) { std::memcpy(q, "INF", 3); return (int)(q - o) + 3; }
    if (v == 0.0) { std::memcpy(q, "0.0000E+00", 10); return (int)(q - o) + 10; }
    int ex;
    std::frexp(v, &ex);
    int e = (int)std::floor((ex - 1) * 0.30102999566398120); // within one of floor(log10 v); corrected below
    for (int attempt = 0; attempt < 3; ++attempt) {
        const int k = e - 4; // scaled = v / 10^k should land in [1e4, 1e5)
        if (k < -22 || k > 22) break;
        const double s = k >= 0 ? v / P10[k] : v * P10[-k];
        if (s >= 100000.0) { ++e; continue; }
        if (s < 10000.0) { --e; continue; }
        double m = std::floor(s);
        const double frac = s - m;
        if (std::fabs(frac - 0.5) < 1e-6) break; // too close to a tie to trust one rounded operation
        if (frac > 0.5) m += 1.0;
        long mi = (long)m;
        if (mi == 100000) { mi = 10000; ++e; }
        q[0] = (char)('0' + mi / 10000); q[1] = '.';
        q[2] = (char)('0' + (mi / 1000) % 10); q[3] = (char)('0' + (mi / 100) % 10);
        q[4] = (char)('0' + (mi / 10) % 10); q[5] = (char)('0' + mi % 10);
        q[6] = 'E'; q[7] = e < 0 ? '-' : '+';
        const int ae = e < 0 ? -e : e;
        if (ae >= 100) { q[8] = (char)('0' + ae / 100); q[9] = (char)('0' + (ae / 10) % 10); q[10] = (char)('0' + ae % 10); return (int)(q - o) + 11; }
        q[8] = (char)('0' + ae / 10); q[9] = (char)('0' + ae % 10);
        return (int)(q - o) + 10;
    }
    return (int)(q - o) + std::snprintf(q, 32, "%.4E", v);
}

inline int format_thrm(const double *row17, char *out)
{
    char *q = out;
    for (int c = 0; c < 17; ++c) { *q++ = ' '; q += fmt_e4(row17[c], q); }
    *q++ = '\n';
    return (int)(q - out);
}

inline int format_traj(int natoms, double box, const double *x, char *out)
{
    char *q = out;
    q += std::snprintf(q, 16, "%d ", natoms);
    q += fmt_e4(box, q);
    *q++ = '\n';
    for (int i = 0; i < natoms; ++i) {
        for (int c = 0; c < 3; ++c) { *q++ = ' '; q += fmt_e4(x[3 * i + c], q); }
        *q++ = '\n';
    }
    return (int)(q - out);
}

} // namespace nm
