// nm_lattice.h — init_sample (remcmc:394-433) on the host, for callers of include/nm.h that have no Python: fcc lattice in LAMMPS
// create_atoms order, static relaxation of the box edge to the target pressure (what `fix box/relax iso P` + `minimize` converge to
// for the perfect crystal, remcmc:402-405), uniform random displacement of amplitude DX*LAT (remcmc:407), optional volume
// interpolation of the -is branch (remcmc:409-419; its 1024 NVE steps are nm_run_md).  Runs once per replica, not in the sweep
// loop; plain C++.  Same numbers as neuralmelting_amd/lattice.py (the displacement draws follow numpy's Philox4x64-10 generator
// keyed [seed, global slot], so both front ends start the same chains; tests/test_lattice_abi.py).
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace nm {
namespace lat {

// numpy.random.Philox (Philox4x64-10): counter incremented before each block of four outputs, outputs in order
struct Philox4x64 {
    uint64_t ctr[4] = { 0, 0, 0, 0 }, key[2], buf[4];
    int pos = 4;
    Philox4x64(uint64_t k0, uint64_t k1) { key[0] = k0; key[1] = k1; }
    static void mulhilo(uint64_t a, uint64_t b, uint64_t &hi, uint64_t &lo)
    {
        const unsigned __int128 p = (unsigned __int128)a * b;
        hi = (uint64_t)(p >> 64); lo = (uint64_t)p;
    }
    void block()
    {
        if (++ctr[0] == 0 && ++ctr[1] == 0 && ++ctr[2] == 0) ++ctr[3];
        uint64_t c[4] = { ctr[0], ctr[1], ctr[2], ctr[3] }, k[2] = { key[0], key[1] };
        for (int r = 0; r < 10; ++r) {
            if (r) { k[0] += 0x9E3779B97F4A7C15ull; k[1] += 0xBB67AE8584CAA73Bull; }
            uint64_t hi0, lo0, hi1, lo1;
            mulhilo(0xD2E7470EE14C6C93ull, c[0], hi0, lo0);
            mulhilo(0xCA5A826395121157ull, c[2], hi1, lo1);
            const uint64_t n0 = hi1 ^ c[1] ^ k[0], n2 = hi0 ^ c[3] ^ k[1];
            c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        }
        for (int q = 0; q < 4; ++q) buf[q] = c[q];
        pos = 0;
    }
    double uniform() // Generator.random(): 53 bits
    {
        if (pos >= 4) block();
        return (double)(buf[pos++] >> 11) * (1.0 / 9007199254740992.0);
    }
};

// fractional coordinates in create_atoms order: k outer, j, i inner, basis innermost
inline void fcc_fractional(int sz, std::vector<double> &f)
{
    static const double B[4][3] = { { 0, 0, 0 }, { 0.5, 0.5, 0 }, { 0.5, 0, 0.5 }, { 0, 0.5, 0.5 } };
    f.resize((size_t)12 * sz * sz * sz);
    size_t n = 0;
    for (int k = 0; k < sz; ++k)
        for (int j = 0; j < sz; ++j)
            for (int i = 0; i < sz; ++i)
                for (int b = 0; b < 4; ++b) {
                    f[n++] = (i + B[b][0]) / sz; f[n++] = (j + B[b][1]) / sz; f[n++] = (k + B[b][2]) / sz;
                }
}

// static virial pressure W / 3V of the perfect 4^3 crystal at box edge `box` (minimum image: 4^3 cells exceed twice either cutoff)
inline double static_pressure(int element, const std::vector<double> &f, double box)
{
    const int n = (int)(f.size() / 3);
    double w = 0.0;
    if (element == 0) { // lj/cut 2.5, unshifted: sum over pairs of r.f = 48 r^-12 - 24 r^-6
        for (int a = 0; a < n; ++a)
            for (int b = a + 1; b < n; ++b) {
                double r2 = 0.0;
                for (int c = 0; c < 3; ++c) { double d = f[3 * a + c] - f[3 * b + c]; d -= std::nearbyint(d); r2 += d * d; }
                r2 *= box * box;
                if (r2 < 6.25) { const double r6i = 1.0 / (r2 * r2 * r2); w += r6i * (48.0 * r6i - 24.0); }
            }
        return w / (3.0 * box * box * box);
    }
    // Sutton-Chen Al (nm_api.hip fill_params): densities, then the pair part of r.f; returned in bar (LAMMPS metal units)
    const double eps = 0.033147, a2 = 4.05 * 4.05, cc = 16.399, rc2 = 7.5 * 7.5;
    std::vector<double> rho((size_t)n, 0.0);
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b) {
            if (a == b) continue;
            double r2 = 0.0;
            for (int c = 0; c < 3; ++c) { double d = f[3 * a + c] - f[3 * b + c]; d -= std::nearbyint(d); r2 += d * d; }
            r2 *= box * box;
            if (r2 < rc2) { const double q2 = a2 / r2; rho[a] += q2 * q2 * q2; }
        }
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b) {
            if (a == b) continue;
            double r2 = 0.0;
            for (int c = 0; c < 3; ++c) { double d = f[3 * a + c] - f[3 * b + c]; d -= std::nearbyint(d); r2 += d * d; }
            r2 *= box * box;
            if (r2 < rc2) {
                const double q2 = a2 / r2, rm = q2 * q2 * q2, rn = rm * std::sqrt(q2);
                const double dF = 0.5 * cc * (1.0 / std::sqrt(rho[a]) + 1.0 / std::sqrt(rho[b]));
                w += 0.5 * eps * (7.0 * rn - 6.0 * dF * rm);
            }
        }
    return w / (3.0 * box * box * box) * 1.6021765e6;
}

// box edge of the sz^3 crystal at which the static pressure equals `press`: bracket, then bisection refined by secant steps
// (the pressure is smooth and monotonic in the edge); converged far below the '%f' resolution the engine rounds the box to
inline double relax_box(int element, int sz, double press)
{
    std::vector<double> f;
    fcc_fractional(4, f); // the perfect lattice's pressure depends on the lattice constant only: 4^3 root, scaled
    const double a0 = 4.0 * (element == 0 ? std::cbrt(4.0 / 1.122) : 4.046);
    auto g = [&](double b) { return static_pressure(element, f, b) - press; };
    double lo = (element == 0 ? 0.9 : 0.97) * a0, hi = (element == 0 ? 1.05 : 1.03) * a0;
    while (g(lo) < 0.0) lo *= (element == 0 ? 0.97 : 0.99);
    while (g(hi) > 0.0) hi *= (element == 0 ? 1.02 : 1.01);
    double glo = g(lo), ghi = g(hi);
    for (int it = 0; it < 200 && (hi - lo) > 1e-14 * hi; ++it) {
        double m = lo - glo * (hi - lo) / (ghi - glo);            // secant
        if (!(m > lo && m < hi) || (it & 3) == 3) m = 0.5 * (lo + hi); // ... kept honest by a bisection every fourth step
        const double gm = g(m);
        if (gm == 0.0) { lo = hi = m; break; }
        if (gm > 0.0) { lo = m; glo = gm; } else { hi = m; ghi = gm; }
    }
    return 0.5 * (lo + hi) * (sz / 4.0);
}

// state of global slot k = i*nt + j: x[3N] (wrapped into the box), box edge.  box_row = relax_box of pressure row i.
inline void init_state(int element, int sz, double box_row, uint32_t seed, int gslot, int j, int nt, double dx, int interpolate,
                       const std::vector<double> &frac, double *x, double *box)
{
    const int n = (int)(frac.size() / 3);
    const double amp = dx * (element == 0 ? 1.122 : 4.046);
    Philox4x64 rng((uint64_t)seed, (uint64_t)gslot);
    double bk = box_row;
    for (int a = 0; a < 3 * n; ++a) {
        double v = frac[a] * box_row + amp * 2.0 * (rng.uniform() - 0.5);
        v -= std::floor(v / box_row) * box_row;
        x[a] = v;
    }
    if (interpolate) { // -is (remcmc:409-419): expand the volume by exp(0.75 (j+1)/NT) about the origin
        bk = std::cbrt(std::exp(std::log(box_row * box_row * box_row) + 0.75 * (j + 1) / nt));
        for (int a = 0; a < 3 * n; ++a) x[a] = x[a] * (bk / box_row);
    }
    *box = bk;
}

} // namespace lat
} // namespace nm
