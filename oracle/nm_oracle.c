/*
 * nm_oracle.c — CPU ORACLE (test infrastructure, NOT product code).  See nm_oracle.h.
 *
 * Every function cites the reference lines it restates: "remcmc" =
 * /root/reference/scripts/lammps_remcmc.py.  LAMMPS behaviour is restated from its published
 * algorithm (pair_lj_cut, fix_nve, velocity, displace_atoms, compute_pe/ke/temp/pressure);
 * liblammps itself is not available in this image ("parity unpinned" for that arithmetic).
 */
#include "nm_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ constants */
#define RC 2.5    /* pair_style lj/cut 2.5, remcmc:365 */
#define SKIN 0.3  /* LAMMPS lj-units default neighbour skin (not observable in results) */

enum { S_ROLL = 0, S_ACC = 1, S_VOL = 2, S_DISP_XY = 3, S_DISP_Z = 4, S_VEL_A = 5, S_VEL_B = 6,
       S_EXCH = 7, S_ITER_XY = 8, S_ITER_Z = 9, S_ITER_ACC = 10 };

/* Sutton-Chen Al (A. P. Sutton, J. Chen, Phil. Mag. Lett. 61 (1990) 139): the build's own choice for
   the non-LJ config (the reference's MEAM files are absent: SURVEY.md §8 a-10). */
#define SC_EPS 0.033147
#define SC_A 4.05
#define SC_C 16.399
#define SC_N 7
#define SC_M 6
#define SC_RC 7.5

struct orc_sim {
    int n, units, pot;
    double mass;
    double kB, mvv2e, ftm2v, nktv2p;
    double rc, skin;
    double L, h;
    double *x, *v, *f;
    int *img;
    double U, W;
    int npairs;
    /* half Verlet list */
    int *nstart, *nj;
    int ncap;
    double *x0;
    double L0;
    int list_ok;
    int fresh; /* U/W/f correspond to the current x and L */
    double *rho; /* EAM density */
    uint32_t seed, slot, step;
};

/* ------------------------------------------------------------------ Philox4x32-10 */
static inline void mulhilo(uint32_t a, uint32_t b, uint32_t *hi, uint32_t *lo)
{
    uint64_t p = (uint64_t)a * (uint64_t)b;
    *hi = (uint32_t)(p >> 32);
    *lo = (uint32_t)p;
}

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        mulhilo(0xD2511F53u, c0, &hi0, &lo0);
        mulhilo(0xCD9E8D57u, c2, &hi1, &lo1);
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n1 = lo1;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        uint32_t n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

double orc_u01(uint32_t hi, uint32_t lo)
{
    uint64_t w = ((uint64_t)hi << 32) | (uint64_t)lo;
    return (double)(w >> 11) * (1.0 / 9007199254740992.0); /* 2^-53 → [0,1) */
}

static void rng4(const orc_sim *s, uint32_t index, uint32_t stream, uint32_t tag, uint32_t out[4])
{
    uint32_t ctr[4] = { index, stream, tag, s->step };
    uint32_t key[2] = { s->seed, s->slot };
    orc_philox4x32_10(ctr, key, out);
}

/* ------------------------------------------------------------------ '%f' round trip */
__attribute__((optimize("fp-contract=off"))) double orc_q6(double x)
{
    /* value of float('%f' % x): nearest double to the 6-decimal correctly rounded decimal of x */
    double p = x * 1.0e6;
    double n = rint(p);
    if (fabs(p - n) == 0.5) { /* fl(x*1e6) sits on a tie: decide with the exact residual of the product */
        double e = fma(x, 1.0e6, -p);
        if (e > 0.0) n = floor(p) + 1.0;
        else if (e < 0.0) n = floor(p);
        /* e == 0: exact tie, printf rounds half to even = rint */
    }
    return n / 1.0e6;
}

double orc_q6_printf(double x)
{
    char buf[512];
    snprintf(buf, sizeof buf, "%f", x);
    return strtod(buf, NULL);
}

/* ------------------------------------------------------------------ life cycle */
orc_sim *orc_create(int natoms, int units, double mass, int pot)
{
    orc_sim *s = (orc_sim *)calloc(1, sizeof *s);
    if (!s) return NULL;
    s->n = natoms; s->units = units; s->pot = pot; s->mass = mass;
    if (units == 0) { s->kB = 1.0; s->mvv2e = 1.0; s->ftm2v = 1.0; s->nktv2p = 1.0; }
    else { /* LAMMPS metal units */
        s->kB = 8.617343e-5; s->mvv2e = 1.0364269e-4; s->ftm2v = 1.0 / 1.0364269e-4; s->nktv2p = 1.6021765e6;
    }
    s->rc = (pot == 0) ? RC : SC_RC;
    s->skin = (units == 0) ? SKIN : 2.0; /* metal-units default skin 2.0 A */
    s->x = (double *)calloc(3 * (size_t)natoms, sizeof(double));
    s->v = (double *)calloc(3 * (size_t)natoms, sizeof(double));
    s->f = (double *)calloc(3 * (size_t)natoms, sizeof(double));
    s->x0 = (double *)calloc(3 * (size_t)natoms, sizeof(double));
    s->rho = (double *)calloc((size_t)natoms, sizeof(double));
    s->img = (int *)calloc(3 * (size_t)natoms, sizeof(int));
    s->nstart = (int *)calloc((size_t)natoms + 1, sizeof(int));
    s->ncap = natoms * 96;
    s->nj = (int *)malloc((size_t)s->ncap * sizeof(int));
    s->L = 0.0; s->h = 0.0; s->list_ok = 0;
    return s;
}

void orc_destroy(orc_sim *s)
{
    if (!s) return;
    free(s->x); free(s->v); free(s->f); free(s->x0); free(s->rho); free(s->img); free(s->nstart); free(s->nj);
    free(s);
}

void orc_set_rng(orc_sim *s, uint32_t seed, uint32_t slot, uint32_t step) { s->seed = seed; s->slot = slot; s->step = step; }
void orc_set_box(orc_sim *s, double L) { s->L = L; s->fresh = 0; }
double orc_get_box(const orc_sim *s) { return s->L; }
void orc_set_x(orc_sim *s, const double *x) { memcpy(s->x, x, 3 * (size_t)s->n * sizeof(double)); s->fresh = 0; }
void orc_set_v(orc_sim *s, const double *v) { memcpy(s->v, v, 3 * (size_t)s->n * sizeof(double)); }
void orc_get_x(const orc_sim *s, double *x) { memcpy(x, s->x, 3 * (size_t)s->n * sizeof(double)); }
void orc_get_v(const orc_sim *s, double *v) { memcpy(v, s->v, 3 * (size_t)s->n * sizeof(double)); }
void orc_get_f(const orc_sim *s, double *f) { memcpy(f, s->f, 3 * (size_t)s->n * sizeof(double)); }
void orc_get_image(const orc_sim *s, int *img) { memcpy(img, s->img, 3 * (size_t)s->n * sizeof(int)); }
void orc_set_image(orc_sim *s, const int *img) { memcpy(s->img, img, 3 * (size_t)s->n * sizeof(int)); }
void orc_set_timestep(orc_sim *s, double h) { s->h = h; }
double orc_pe(const orc_sim *s) { return s->U; }
double orc_virial(const orc_sim *s) { return s->W; }
int orc_nlist(const orc_sim *s) { return s->nstart[s->n]; }
int orc_npairs(const orc_sim *s) { return s->npairs; }

/* ------------------------------------------------------------------ wrap (domain->remap) */
static void wrap_all(orc_sim *s)
{
    const double L = s->L;
    for (int a = 0; a < 3 * s->n; ++a) {
        double x = s->x[a];
        if (x < 0.0 || x >= L) {
            double nb = floor(x / L);
            x -= nb * L;
            int im = (int)nb;
            if (x >= L) { x -= L; im += 1; }
            if (x < 0.0) x = 0.0;
            s->x[a] = x;
            s->img[a] += im;
        }
    }
}

/* ------------------------------------------------------------------ neighbour list */
static inline double minimg(double d, double L, double invL) { return d - L * rint(d * invL); }

static int list_valid(const orc_sim *s)
{
    if (!s->list_ok) return 0;
    const double L = s->L, invL = 1.0 / L, sc = L / s->L0;
    const double thr = 0.5 * (sc * (s->rc + s->skin) - s->rc);
    if (!(thr > 0.0)) return 0;
    const double thr2 = thr * thr;
    for (int i = 0; i < s->n; ++i) {
        double dx = minimg(s->x[3 * i] - sc * s->x0[3 * i], L, invL);
        double dy = minimg(s->x[3 * i + 1] - sc * s->x0[3 * i + 1], L, invL);
        double dz = minimg(s->x[3 * i + 2] - sc * s->x0[3 * i + 2], L, invL);
        if (dx * dx + dy * dy + dz * dz > thr2) return 0;
    }
    return 1;
}

static int list_build(orc_sim *s)
{
    const double L = s->L, invL = 1.0 / L;
    const double rl = s->rc + s->skin, rl2 = rl * rl;
    int cnt = 0;
    for (int i = 0; i < s->n; ++i) {
        s->nstart[i] = cnt;
        const double xi = s->x[3 * i], yi = s->x[3 * i + 1], zi = s->x[3 * i + 2];
        for (int j = i + 1; j < s->n; ++j) {
            double dx = minimg(xi - s->x[3 * j], L, invL);
            double dy = minimg(yi - s->x[3 * j + 1], L, invL);
            double dz = minimg(zi - s->x[3 * j + 2], L, invL);
            if (dx * dx + dy * dy + dz * dz < rl2) {
                if (cnt == s->ncap) {
                    s->ncap *= 2;
                    s->nj = (int *)realloc(s->nj, (size_t)s->ncap * sizeof(int));
                    if (!s->nj) return -1;
                }
                s->nj[cnt++] = j;
            }
        }
    }
    s->nstart[s->n] = cnt;
    memcpy(s->x0, s->x, 3 * (size_t)s->n * sizeof(double));
    s->L0 = L;
    s->list_ok = 1;
    return 0;
}

/* ------------------------------------------------------------------ pair evaluation */
/* lj/cut: pair_lj_cut.cpp compute(): r2inv, r6inv, forcelj = r6inv*(48 r6inv - 24), fpair = forcelj*r2inv,
   evdwl = r6inv*(4 r6inv - 4); no shift, no tail (pair_modify defaults; deck remcmc:364-367). */
static void eval_lj(orc_sim *s)
{
    const double L = s->L, invL = 1.0 / L, rc2 = s->rc * s->rc;
    double U = 0.0, W = 0.0;
    int np = 0;
    memset(s->f, 0, 3 * (size_t)s->n * sizeof(double));
    for (int i = 0; i < s->n; ++i) {
        const double xi = s->x[3 * i], yi = s->x[3 * i + 1], zi = s->x[3 * i + 2];
        double fx = 0.0, fy = 0.0, fz = 0.0;
        for (int q = s->nstart[i]; q < s->nstart[i + 1]; ++q) {
            const int j = s->nj[q];
            double dx = minimg(xi - s->x[3 * j], L, invL);
            double dy = minimg(yi - s->x[3 * j + 1], L, invL);
            double dz = minimg(zi - s->x[3 * j + 2], L, invL);
            double r2 = dx * dx + dy * dy + dz * dz;
            if (r2 < rc2) {
                double r2i = 1.0 / r2;
                double r6i = r2i * r2i * r2i;
                double fp = r6i * (48.0 * r6i - 24.0) * r2i;
                fx += dx * fp; fy += dy * fp; fz += dz * fp;
                s->f[3 * j] -= dx * fp; s->f[3 * j + 1] -= dy * fp; s->f[3 * j + 2] -= dz * fp;
                U += r6i * (4.0 * r6i - 4.0);
                W += r2 * fp;
                ++np;
            }
        }
        s->f[3 * i] += fx; s->f[3 * i + 1] += fy; s->f[3 * i + 2] += fz;
    }
    s->U = U; s->W = W; s->npairs = np;
}

/* Sutton-Chen EAM: E = eps * [ 1/2 sum_ij (a/r)^n - c sum_i sqrt(rho_i) ], rho_i = sum_j (a/r)^m, r < 2a */
static void eval_sc(orc_sim *s)
{
    const double L = s->L, invL = 1.0 / L, rc2 = s->rc * s->rc;
    double U = 0.0, W = 0.0;
    int np = 0;
    memset(s->f, 0, 3 * (size_t)s->n * sizeof(double));
    memset(s->rho, 0, (size_t)s->n * sizeof(double));
    for (int i = 0; i < s->n; ++i)
        for (int q = s->nstart[i]; q < s->nstart[i + 1]; ++q) {
            const int j = s->nj[q];
            double dx = minimg(s->x[3 * i] - s->x[3 * j], L, invL);
            double dy = minimg(s->x[3 * i + 1] - s->x[3 * j + 1], L, invL);
            double dz = minimg(s->x[3 * i + 2] - s->x[3 * j + 2], L, invL);
            double r2 = dx * dx + dy * dy + dz * dz;
            if (r2 < rc2) {
                double q2 = (SC_A * SC_A) / r2;
                double rm = q2 * q2 * q2; /* (a/r)^6 */
                s->rho[i] += rm; s->rho[j] += rm;
            }
        }
    for (int i = 0; i < s->n; ++i) U -= SC_EPS * SC_C * sqrt(s->rho[i]);
    for (int i = 0; i < s->n; ++i)
        for (int q = s->nstart[i]; q < s->nstart[i + 1]; ++q) {
            const int j = s->nj[q];
            double dx = minimg(s->x[3 * i] - s->x[3 * j], L, invL);
            double dy = minimg(s->x[3 * i + 1] - s->x[3 * j + 1], L, invL);
            double dz = minimg(s->x[3 * i + 2] - s->x[3 * j + 2], L, invL);
            double r2 = dx * dx + dy * dy + dz * dz;
            if (r2 < rc2) {
                double r2i = 1.0 / r2;
                double q2 = (SC_A * SC_A) * r2i;
                double rm = q2 * q2 * q2;        /* (a/r)^6 */
                double rn = rm * sqrt(q2);       /* (a/r)^7 */
                double dF = 0.5 * SC_C * (1.0 / sqrt(s->rho[i]) + 1.0 / sqrt(s->rho[j]));
                /* -dE/dr / r = eps * [ n (a/r)^n - m dF (a/r)^m ] / r^2 */
                double fp = SC_EPS * (SC_N * rn - SC_M * dF * rm) * r2i;
                s->f[3 * i] += dx * fp; s->f[3 * i + 1] += dy * fp; s->f[3 * i + 2] += dz * fp;
                s->f[3 * j] -= dx * fp; s->f[3 * j + 1] -= dy * fp; s->f[3 * j + 2] -= dz * fp;
                U += SC_EPS * rn;
                W += r2 * fp;
                ++np;
            }
        }
    s->U = U; s->W = W; s->npairs = np;
}

static int eval_current(orc_sim *s)
{
    if (!(s->L >= 2.0 * s->rc)) return -2; /* minimum-image limit (LAMMPS would use ghost images) */
    if (!list_valid(s)) { if (list_build(s)) return -1; }
    if (s->pot == 0) eval_lj(s); else eval_sc(s);
    s->fresh = 1;
    return 0;
}

/* "run 0" (remcmc:469,484,496,...): full setup = remap atoms into the box, re-neighbour, forces, thermo */
int orc_setup(orc_sim *s)
{
    wrap_all(s);
    if (s->fresh) return 0; /* nothing moved since the last evaluation: same U, W, f */
    return eval_current(s);
}

/* O(N^2) direct sum, independent of the list machinery (LJ only) */
int orc_eval_allpairs(orc_sim *s, double *Uo, double *Wo, double *f)
{
    const double L = s->L, invL = 1.0 / L, rc2 = s->rc * s->rc;
    double U = 0.0, W = 0.0;
    if (f) memset(f, 0, 3 * (size_t)s->n * sizeof(double));
    if (s->pot != 0) return -1;
    for (int i = 0; i < s->n; ++i)
        for (int j = i + 1; j < s->n; ++j) {
            double dx = minimg(s->x[3 * i] - s->x[3 * j], L, invL);
            double dy = minimg(s->x[3 * i + 1] - s->x[3 * j + 1], L, invL);
            double dz = minimg(s->x[3 * i + 2] - s->x[3 * j + 2], L, invL);
            double r2 = dx * dx + dy * dy + dz * dz;
            if (r2 < rc2) {
                double r2i = 1.0 / r2, r6i = r2i * r2i * r2i;
                double fp = r6i * (48.0 * r6i - 24.0) * r2i;
                U += r6i * (4.0 * r6i - 4.0);
                W += r2 * fp;
                if (f) {
                    f[3 * i] += dx * fp; f[3 * i + 1] += dy * fp; f[3 * i + 2] += dz * fp;
                    f[3 * j] -= dx * fp; f[3 * j + 1] -= dy * fp; f[3 * j + 2] -= dz * fp;
                }
            }
        }
    *Uo = U; *Wo = W;
    return 0;
}

/* ------------------------------------------------------------------ thermo computes */
static double sum_mv2(const orc_sim *s)
{
    double t = 0.0;
    for (int i = 0; i < s->n; ++i)
        t += s->mass * (s->v[3 * i] * s->v[3 * i] + s->v[3 * i + 1] * s->v[3 * i + 1] + s->v[3 * i + 2] * s->v[3 * i + 2]);
    return t;
}
double orc_ke(const orc_sim *s) { return 0.5 * s->mvv2e * sum_mv2(s); }                       /* compute ke */
double orc_temp(const orc_sim *s) { double dof = 3.0 * s->n - 3.0; return sum_mv2(s) * s->mvv2e / (dof * s->kB); } /* compute temp */
double orc_press(const orc_sim *s)                                                              /* compute pressure */
{
    double dof = 3.0 * s->n - 3.0;
    double vol = s->L * s->L * s->L;
    return (dof * s->kB * orc_temp(s) + s->W) / 3.0 * (1.0 / vol) * s->nktv2p;
}

/* ------------------------------------------------------------------ moves' LAMMPS commands */
/* displace_atoms all random a a a seed units box (remcmc:483): x += a*2*(u-0.5) per axis, then remap */
void orc_displace(orc_sim *s, double a, uint32_t tag)
{
    for (int i = 0; i < s->n; ++i) {
        uint32_t o[4], p[4];
        rng4(s, (uint32_t)i, S_DISP_XY, tag, o);
        rng4(s, (uint32_t)i, S_DISP_Z, tag, p);
        s->x[3 * i] += a * 2.0 * (orc_u01(o[0], o[1]) - 0.5);
        s->x[3 * i + 1] += a * 2.0 * (orc_u01(o[2], o[3]) - 0.5);
        s->x[3 * i + 2] += a * 2.0 * (orc_u01(p[0], p[1]) - 0.5);
    }
    s->fresh = 0;
    wrap_all(s);
}

static void vcm_of(const orc_sim *s, double vcm[3])
{
    double p[3] = { 0, 0, 0 };
    for (int i = 0; i < s->n; ++i) for (int c = 0; c < 3; ++c) p[c] += s->mass * s->v[3 * i + c];
    double mt = s->mass * s->n;
    for (int c = 0; c < 3; ++c) vcm[c] = p[c] / mt;
}

void orc_zero_linear(orc_sim *s)
{
    double vcm[3];
    vcm_of(s, vcm);
    for (int i = 0; i < s->n; ++i) for (int c = 0; c < 3; ++c) s->v[3 * i + c] -= vcm[c];
}

/* velocity.cpp zero_rotation(): xcm/angmom/inertia on unwrapped coordinates, omega = I^-1 L, v -= omega x r */
void orc_zero_angular(orc_sim *s)
{
    const double L = s->L, m = s->mass;
    double xcm[3] = { 0, 0, 0 };
    for (int i = 0; i < s->n; ++i) for (int c = 0; c < 3; ++c) xcm[c] += m * (s->x[3 * i + c] + s->img[3 * i + c] * L);
    for (int c = 0; c < 3; ++c) xcm[c] /= m * s->n;
    double am[3] = { 0, 0, 0 }, I[3][3] = { { 0 } };
    for (int i = 0; i < s->n; ++i) {
        double dx = s->x[3 * i] + s->img[3 * i] * L - xcm[0];
        double dy = s->x[3 * i + 1] + s->img[3 * i + 1] * L - xcm[1];
        double dz = s->x[3 * i + 2] + s->img[3 * i + 2] * L - xcm[2];
        const double *v = &s->v[3 * i];
        am[0] += m * (dy * v[2] - dz * v[1]);
        am[1] += m * (dz * v[0] - dx * v[2]);
        am[2] += m * (dx * v[1] - dy * v[0]);
        I[0][0] += m * (dy * dy + dz * dz);
        I[1][1] += m * (dx * dx + dz * dz);
        I[2][2] += m * (dx * dx + dy * dy);
        I[0][1] -= m * dx * dy;
        I[1][2] -= m * dy * dz;
        I[0][2] -= m * dx * dz;
    }
    I[1][0] = I[0][1]; I[2][1] = I[1][2]; I[2][0] = I[0][2];
    double det = I[0][0] * I[1][1] * I[2][2] + I[0][1] * I[1][2] * I[2][0] + I[0][2] * I[1][0] * I[2][1]
               - I[0][0] * I[1][2] * I[2][1] - I[0][1] * I[1][0] * I[2][2] - I[2][0] * I[1][1] * I[0][2];
    double w[3] = { 0, 0, 0 };
    if (det > 0.0) {
        double inv[3][3];
        inv[0][0] = I[1][1] * I[2][2] - I[1][2] * I[2][1];
        inv[0][1] = -(I[0][1] * I[2][2] - I[0][2] * I[2][1]);
        inv[0][2] = I[0][1] * I[1][2] - I[0][2] * I[1][1];
        inv[1][0] = -(I[1][0] * I[2][2] - I[1][2] * I[2][0]);
        inv[1][1] = I[0][0] * I[2][2] - I[0][2] * I[2][0];
        inv[1][2] = -(I[0][0] * I[1][2] - I[0][2] * I[1][0]);
        inv[2][0] = I[1][0] * I[2][1] - I[1][1] * I[2][0];
        inv[2][1] = -(I[0][0] * I[2][1] - I[0][1] * I[2][0]);
        inv[2][2] = I[0][0] * I[1][1] - I[0][1] * I[1][0];
        for (int a = 0; a < 3; ++a)
            w[a] = (inv[a][0] * am[0] + inv[a][1] * am[1] + inv[a][2] * am[2]) / det;
    }
    for (int i = 0; i < s->n; ++i) {
        double dx = s->x[3 * i] + s->img[3 * i] * L - xcm[0];
        double dy = s->x[3 * i + 1] + s->img[3 * i + 1] * L - xcm[1];
        double dz = s->x[3 * i + 2] + s->img[3 * i + 2] * L - xcm[2];
        s->v[3 * i] -= w[1] * dz - w[2] * dy;
        s->v[3 * i + 1] -= w[2] * dx - w[0] * dz;
        s->v[3 * i + 2] -= w[0] * dy - w[1] * dx;
    }
}

/* velocity all create t seed dist gaussian (remcmc:604): gaussian/sqrt(m) by atom id, remove COM momentum
   (mom yes), rescale to exactly t with dof = 3N-3 (velocity.cpp create()). */
void orc_velocity_create(orc_sim *s, double t, uint32_t tag)
{
    const double twopi = 6.283185307179586476925286766559;
    const double fac = 1.0 / sqrt(s->mass);
    for (int i = 0; i < s->n; ++i) {
        uint32_t o[4], p[4];
        rng4(s, (uint32_t)i, S_VEL_A, tag, o);
        rng4(s, (uint32_t)i, S_VEL_B, tag, p);
        double u1 = orc_u01(o[0], o[1]), u2 = orc_u01(o[2], o[3]);
        double u3 = orc_u01(p[0], p[1]), u4 = orc_u01(p[2], p[3]);
        double r1 = sqrt(-2.0 * log(1.0 - u1)), r2 = sqrt(-2.0 * log(1.0 - u3));
        s->v[3 * i] = r1 * cos(twopi * u2) * fac;
        s->v[3 * i + 1] = r1 * sin(twopi * u2) * fac;
        s->v[3 * i + 2] = r2 * cos(twopi * u4) * fac;
    }
    orc_zero_linear(s);
    double tcur = orc_temp(s);
    double sc = sqrt(t / tcur);
    for (int a = 0; a < 3 * s->n; ++a) s->v[a] *= sc;
}

/* run N (remcmc:616): setup as run 0, then fix nve velocity-Verlet (fix_nve.cpp initial/final_integrate) */
int orc_run(orc_sim *s, int nsteps)
{
    int rc = orc_setup(s);
    if (rc) return rc;
    const double dtf = 0.5 * s->h * s->ftm2v, dtfm = dtf / s->mass, dtv = s->h;
    for (int st = 0; st < nsteps; ++st) {
        for (int a = 0; a < 3 * s->n; ++a) { s->v[a] += dtfm * s->f[a]; s->x[a] += dtv * s->v[a]; }
        rc = eval_current(s); /* no re-wrap inside the run (neigh delay 10 > NSTPS) */
        if (rc) return rc;
        for (int a = 0; a < 3 * s->n; ++a) s->v[a] += dtfm * s->f[a];
    }
    return 0;
}

/* ------------------------------------------------------------------ Metropolis helpers */
typedef struct { const orc_block_params *p; int pos; orc_sim *s; } drawctx;

static double draw_scalar(drawctx *d, uint32_t stream, uint32_t m, uint32_t index)
{
    if (d->p->tape) {
        if (d->pos >= d->p->tape_len) { d->pos++; return 2.0; } /* exhausted: flagged by caller */
        return d->p->tape[d->pos++];
    }
    uint32_t o[4];
    rng4(d->s, index, stream, m, o);
    return orc_u01(o[0], o[1]);
}

static uint32_t draw_tag(drawctx *d, uint32_t m)
{
    if (d->p->tape) { /* np.random.randint(1, 2**16) site: remcmc:482, 603 */
        if (d->pos >= d->p->tape_len) { d->pos++; return 0; }
        return (uint32_t)(d->p->tape[d->pos++] * 65536.0); /* tape stores randint/65536 */
    }
    return m;
}

/* remcmc:487-500 (same at 532-547, 578-593, 623-638): metcrit = exp(-c); +inf → reject without a draw;
   else u = rand(); accept iff u <= min(1, metcrit) with numpy's NaN-propagating min */
static int metropolis(drawctx *d, double c, uint32_t stream, uint32_t m, uint32_t index)
{
    double metcrit = exp(-c);
    if (isinf(metcrit)) return 0;
    double u = draw_scalar(d, stream, m, index);
    double mm = (metcrit != metcrit) ? metcrit : (metcrit < 1.0 ? metcrit : 1.0);
    return (u <= mm) ? 1 : 0;
}

/* ------------------------------------------------------------------ one block (remcmc:665-691) */
int orc_run_block(orc_sim *s, const orc_block_params *p, double *x, double *v, double *box,
                  const double *dxdvdt, double *thermo, double *counters, float *ratios, int *tape_used)
{
    const int n = s->n;
    const double dx = dxdvdt[0], dv = dxdvdt[1], dt = dxdvdt[2];
    double ntp = counters[0], nap = counters[1], ntv = counters[2], nav = counters[3], nth = counters[4], nah = counters[5];
    drawctx d = { p, 0, s };
    int rc;
    double *xs = (double *)malloc(3 * (size_t)n * sizeof(double));
    double *vs = (double *)malloc(3 * (size_t)n * sizeof(double));

    /* init_lammps (remcmc:459-470): new instance, change_box %f, scatter x, v, run 0 */
    memset(s->img, 0, 3 * (size_t)n * sizeof(int));
    s->list_ok = 0;
    orc_set_box(s, orc_q6(*box));
    orc_set_x(s, x);
    orc_set_v(s, v);
    rc = orc_setup(s);
    if (rc) goto done;

    for (int m = 0; m < p->mod; ++m) {
        double roll = draw_scalar(&d, S_ROLL, (uint32_t)m, 0); /* remcmc:645 */
        double branch, crit = 0.0;
        int acc = 0;
        if (roll <= p->ppos && p->bulk) {
            /* bulk_position_mc, remcmc:477-502 */
            branch = 0;
            ntp += 1;
            orc_get_x(s, xs);
            double U0 = s->U, W0 = s->W;
            double pe = s->U / p->et;
            uint32_t tag = draw_tag(&d, (uint32_t)m);
            orc_displace(s, orc_q6(dx * p->lat), tag);
            rc = orc_setup(s);
            if (rc) goto done;
            double penew = s->U / p->et;
            crit = penew - pe;
            acc = metropolis(&d, crit, S_ACC, (uint32_t)m, 0);
            if (acc) nap += 1;
            else {
                orc_set_x(s, xs); /* scatter_atoms restores x only: LAMMPS image flags keep what the remap did */
                wrap_all(s);
                s->U = U0; s->W = W0; /* = result of the reference's re-run "run 0" */
            }
        } else if (roll <= p->ppos) {
            /* iter_position_mc, remcmc:505-549: N single-atom trials, each a full-system evaluation.
               Reference quirk: `od = x[3*k:3*k+3]` (remcmc:522) is a numpy VIEW of x, so after
               `x[3*k:3*k+3] = nd` (remcmc:525) the "revert" at remcmc:540,545 writes nd back onto itself:
               a rejected trial is NOT undone, only not counted.  iter_revert = 0 reproduces that,
               iter_revert = 1 is the corrected move. */
            branch = 3;
            double boxl = orc_get_box(s);
            if (p->iter_revert) wrap_all(s); /* corrected mode: one consistent remap at move start.  Reference mode keeps the
                                                gathered (possibly out-of-box) coordinates on the Python side and re-sends them on
                                                every trial, so LAMMPS remaps them again each time (image flags inflate) */
            orc_get_x(s, xs);
            for (int k = 0; k < n; ++k) {
                ntp += 1;
                double pe = s->U / p->et;
                double od[3] = { xs[3 * k], xs[3 * k + 1], xs[3 * k + 2] };
                double U0 = s->U, W0 = s->W;
                double u3[3];
                if (p->tape) for (int c = 0; c < 3; ++c) u3[c] = draw_scalar(&d, 0, 0, 0); /* rand(3), remcmc:523 */
                else {
                    uint32_t o[4], q[4];
                    rng4(s, (uint32_t)k, S_ITER_XY, (uint32_t)m, o);
                    rng4(s, (uint32_t)k, S_ITER_Z, (uint32_t)m, q);
                    u3[0] = orc_u01(o[0], o[1]); u3[1] = orc_u01(o[2], o[3]); u3[2] = orc_u01(q[0], q[1]);
                }
                for (int c = 0; c < 3; ++c) {
                    double nd = od[c] + 2.0 * (u3[c] - 0.5) * dx * p->lat;   /* no %f here: remcmc:523 */
                    nd -= floor(nd / boxl) * boxl;                          /* remcmc:524 */
                    xs[3 * k + c] = nd;
                }
                orc_set_x(s, xs);
                rc = orc_setup(s);
                if (rc) goto done;
                double penew = s->U / p->et;
                double de = penew - pe;
                int a1 = metropolis(&d, de, S_ITER_ACC, (uint32_t)m, (uint32_t)k);
                if (a1) { nap += 1; acc += 1; }
                else if (!p->iter_revert) {
                    /* remcmc:540-542: the "revert" is a no-op on x (aliasing) but still scatters and runs `run 0`:
                       stale out-of-box coordinates are remapped once more */
                    orc_set_x(s, xs);
                    rc = orc_setup(s);
                    if (rc) goto done;
                } else {
                    xs[3 * k] = od[0]; xs[3 * k + 1] = od[1]; xs[3 * k + 2] = od[2];
                    orc_set_x(s, xs);
                    wrap_all(s);
                    s->U = U0; s->W = W0;
                }
                crit = de;
            }
        } else if (roll <= p->ppos + p->pvol) {
            /* volume_mc, remcmc:552-595 */
            branch = 1;
            ntv += 1;
            double boxl = orc_get_box(s);
            double vol = pow(boxl, 3.0);
            orc_get_x(s, xs);
            double U0 = s->U, W0 = s->W;
            double pe = s->U / p->et;
            double u = draw_scalar(&d, S_VOL, (uint32_t)m, 0);
            double volnew = exp(log(vol) + 2.0 * (u - 0.5) * dv);
            double boxnew = cbrt(volnew);
            double scale = boxnew / boxl;
            for (int a = 0; a < 3 * n; ++a) s->x[a] = scale * xs[a];
            s->fresh = 0;
            orc_set_box(s, orc_q6(boxnew));
            rc = orc_setup(s);
            if (rc) goto done;
            double penew = s->U / p->et;
            crit = (penew - pe) + p->pf * (volnew - vol) - (n + 1) * log(volnew / vol); /* remcmc:576 */
            acc = metropolis(&d, crit, S_ACC, (uint32_t)m, 0);
            if (acc) nav += 1;
            else {
                orc_set_box(s, orc_q6(boxl));
                orc_set_x(s, xs);
                wrap_all(s);
                s->U = U0; s->W = W0;
            }
        } else {
            /* hamiltonian_mc, remcmc:598-640 */
            branch = 2;
            nth += 1;
            uint32_t tag = draw_tag(&d, (uint32_t)m);
            orc_velocity_create(s, orc_q6(p->t), tag);
            orc_zero_linear(s);
            orc_zero_angular(s);
            orc_set_timestep(s, orc_q6(dt));
            rc = orc_setup(s);
            if (rc) goto done;
            orc_get_x(s, xs); orc_get_v(s, vs);
            double U0 = s->U, W0 = s->W;
            double etot = s->U / p->et + orc_ke(s) / p->et;
            rc = orc_run(s, p->nstps);
            if (rc) goto done;
            double etotnew = s->U / p->et + orc_ke(s) / p->et;
            crit = etotnew - etot;
            acc = metropolis(&d, crit, S_ACC, (uint32_t)m, 0);
            if (acc) nah += 1;
            else {
                orc_set_x(s, xs); orc_set_v(s, vs);
                wrap_all(s);
                s->U = U0; s->W = W0;
            }
        }
        if (p->trace) {
            p->trace[4 * m] = branch; p->trace[4 * m + 1] = acc; p->trace[4 * m + 2] = crit; p->trace[4 * m + 3] = s->U;
        }
    }
    /* lammps_extract, remcmc:377-391 */
    orc_get_x(s, x); orc_get_v(s, v);
    *box = orc_get_box(s);
    thermo[0] = orc_temp(s); thermo[1] = orc_pe(s); thermo[2] = orc_ke(s); thermo[3] = orc_press(s);
    thermo[4] = pow(*box, 3.0);
    counters[0] = ntp; counters[1] = nap; counters[2] = ntv; counters[3] = nav; counters[4] = nth; counters[5] = nah;
    /* remcmc:685-688: float32 ratios, 0/0 -> 0 */
    ratios[0] = (ntp > 0) ? (float)nap / (float)ntp : 0.0f;
    ratios[1] = (ntv > 0) ? (float)nav / (float)ntv : 0.0f;
    ratios[2] = (nth > 0) ? (float)nah / (float)nth : 0.0f;
    rc = 0;
    if (p->tape && d.pos > p->tape_len) rc = -3;
done:
    if (tape_used) *tape_used = d.pos;
    free(xs); free(vs);
    return rc;
}

int orc_run_blocks(int ns, int natoms, int units, double mass, int pot, uint32_t seed, uint32_t slot0,
                   uint32_t step, const orc_block_params *p, double *x, double *v, double *box,
                   const double *dxdvdt, double *thermo, double *counters, float *ratios, int nthreads)
{
    int err = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int k = 0; k < ns; ++k) {
        orc_sim *s = orc_create(natoms, units, mass, pot);
        orc_set_rng(s, seed, slot0 + (uint32_t)k, step);
        int rc = orc_run_block(s, &p[k], x + 3 * (size_t)natoms * k, v + 3 * (size_t)natoms * k, box + k,
                               dxdvdt + 3 * k, thermo + 5 * k, counters + 6 * k, ratios + 3 * k, NULL);
        if (rc) {
#ifdef _OPENMP
#pragma omp critical
#endif
            err = rc;
        }
        orc_destroy(s);
    }
    return err;
}

/* gen_mc_param, remcmc:726-745 */
void orc_adapt(const float ratios[3], double dxdvdt[3])
{
    for (int c = 0; c < 3; ++c) {
        if (ratios[c] < 0.5) dxdvdt[c] = 0.9375 * dxdvdt[c];
        if (ratios[c] > 0.5) dxdvdt[c] = 1.0625 * dxdvdt[c];
    }
}

/* replica_exchange, remcmc:776-803 */
int orc_exchange(int np_total, int nt, int row0, int nrows, uint32_t seed, uint32_t step, double *etot,
                 double *vol, const double *et, const double *pf, int *perm, const double *tape, double *crit_out)
{
    (void)np_total;
    int swaps = 0, tp = 0;
    for (int ul = 0; ul < nrows; ++ul) {
        const int u = row0 + ul;
        const int pairs_per_row = nt * (nt - 1) / 2;
        int q = 0;
        for (int vv = nt - 1; vv >= 0; --vv)
            for (int w = 0; w < vv; ++w, ++q) {
                const int i = ul * nt + vv, j = ul * nt + w; /* local slots */
                double de = etot[i] - etot[j];
                double dvv = vol[i] - vol[j];
                double dh = de * (1.0 / et[i] - 1.0 / et[j]) + (pf[i] - pf[j]) * dvv;
                double uu;
                if (tape) uu = tape[tp];
                else {
                    uint32_t ctr[4] = { (uint32_t)(u * pairs_per_row + q), S_EXCH, 0u, step };
                    uint32_t key[2] = { seed, 0xFFFFFFFFu };
                    uint32_t o[4];
                    orc_philox4x32_10(ctr, key, o);
                    uu = orc_u01(o[0], o[1]);
                }
                if (crit_out) crit_out[tp] = dh;
                ++tp;
                double e = exp(dh);
                double mm = (e != e) ? e : (e < 1.0 ? e : 1.0);
                if (uu <= mm) {
                    ++swaps;
                    double t0 = etot[i]; etot[i] = etot[j]; etot[j] = t0;
                    t0 = vol[i]; vol[i] = vol[j]; vol[j] = t0;
                    int b = perm[i]; perm[i] = perm[j]; perm[j] = b;
                }
            }
    }
    return swaps;
}
