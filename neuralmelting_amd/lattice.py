"""Initial-state generation on the host (init_sample, remcmc:394-433) — runs once per replica, not in the
sweep loop: fcc lattice in LAMMPS create_atoms order, static relaxation of the box edge to the target
pressure (what `fix box/relax iso P` + `minimize` converge to for a perfect fcc crystal, remcmc:402-405),
then a uniform random displacement of amplitude DX*LAT (remcmc:407)."""
import numpy as np
from scipy.optimize import brentq

RC = 2.5
# remcmc:873-893
UNITS = {'Ti': 'metal', 'Al': 'metal', 'Ni': 'metal', 'Cu': 'metal', 'LJ': 'lj'}
LAT = {'Ti': ('bcc', 2.951), 'Al': ('fcc', 4.046), 'Ni': ('fcc', 3.524), 'Cu': ('fcc', 3.615), 'LJ': ('fcc', 1.122)}
MASS = {'Ti': 47.867, 'Al': 29.982, 'Ni': 58.693, 'Cu': 63.546, 'LJ': 1.0}
TIMESTEP = {'real': 4.0, 'metal': 0.00390625, 'lj': 0.00390625}

_FCC_BASIS = np.array([[0.0, 0.0, 0.0], [0.5, 0.5, 0.0], [0.5, 0.0, 0.5], [0.0, 0.5, 0.5]])


def lattice_constant(el):
    """`lattice fcc X` (remcmc:340): X is the reduced density in lj units, the cubic cell edge otherwise"""
    if UNITS[el] == 'lj':
        return (4.0 / LAT[el][1]) ** (1.0 / 3.0)
    return LAT[el][1]


def fcc_fractional(sz):
    """fractional coordinates in create_atoms order: k outer, j, i inner, basis innermost"""
    g = np.array([[i, j, k] for k in range(sz) for j in range(sz) for i in range(sz)], dtype=np.float64)
    return ((g[:, None, :] + _FCC_BASIS[None, :, :]) / sz).reshape(-1, 3)


def lj_static(frac, box):
    """U and W = sum r.f of lj/cut 2.5 (unshifted) for fractional coordinates in a cubic box (numpy, O(N^2))"""
    d = frac[:, None, :] - frac[None, :, :]
    d -= np.rint(d)
    r2 = (d * d).sum(-1) * box * box
    iu = np.triu_indices(len(frac), 1)
    r2 = r2[iu]
    r2 = r2[r2 < RC * RC]
    r6i = 1.0 / r2 ** 3
    return float((r6i * (4.0 * r6i - 4.0)).sum()), float((r6i * (48.0 * r6i - 24.0)).sum())


# Sutton-Chen Al (Phil. Mag. Lett. 61 (1990) 139), the engine's EAM for element Al; rc as in the kernels
SC_EPS, SC_A, SC_C, SC_RC = 0.033147, 4.05, 16.399, 7.5
NKTV2P_METAL = 1.6021765e6  # eV/A^3 -> bar (LAMMPS metal units)


def sc_static(frac, box):
    """U [eV] and W = sum r.f [eV] of the Sutton-Chen potential for fractional coordinates in a cubic box (numpy, O(N^2))"""
    d = frac[:, None, :] - frac[None, :, :]
    d -= np.rint(d)
    r2 = (d * d).sum(-1) * box * box
    np.fill_diagonal(r2, np.inf)
    q2 = np.where(r2 < SC_RC * SC_RC, SC_A * SC_A / r2, 0.0)
    rm = q2 ** 3
    rn = rm * np.sqrt(q2)
    rho = rm.sum(1)
    isr = 1.0 / np.sqrt(rho)
    u = SC_EPS * (0.5 * rn.sum() - SC_C * np.sqrt(rho).sum())
    dF = 0.5 * SC_C * (isr[:, None] + isr[None, :])
    w = 0.5 * (SC_EPS * (7.0 * rn - 6.0 * dF * rm)).sum()     # r2*fp summed over pairs once
    return float(u), float(w)


def relax_box(sz, press, el='LJ'):
    """box edge at which the static virial pressure W/(3V) of the perfect lattice equals `press`
    (lj units: reduced pressure; metal units: bar).  The perfect lattice's pressure depends on the lattice constant only, so
    supercells larger than 4^3 (whose box already exceeds twice the cutoff) reuse the 4^3 root: the O(N^2) lattice sum of an
    8^3 cell took 6.5 s per pressure row"""
    if sz > 4:
        return relax_box(4, press, el) * (sz / 4.0)
    frac = fcc_fractional(sz)
    a0 = sz * lattice_constant(el)
    if UNITS[el] == 'metal':
        if el != 'Al':
            raise NotImplementedError('only Al has a potential among the metal-unit elements')

        def f(box):
            return sc_static(frac, box)[1] / (3.0 * box ** 3) * NKTV2P_METAL - press
        lo, hi = 0.97 * a0, 1.03 * a0
        while f(lo) < 0:
            lo *= 0.99
        while f(hi) > 0:
            hi *= 1.01
        return brentq(f, lo, hi, xtol=1e-12, rtol=1e-14)

    def f(box):
        return lj_static(frac, box)[1] / (3.0 * box ** 3) - press
    lo, hi = 0.9 * a0, 1.05 * a0
    while f(lo) < 0:
        lo *= 0.97
    while f(hi) > 0:
        hi *= 1.02
    return brentq(f, lo, hi, xtol=1e-13, rtol=1e-14)


def init_states(sz, P, T, dx, dv, el='LJ', seed=256, row0=0, nrows=None, interpolate=False):
    """STATE for the replicas of pressure rows [row0,row0+nrows): x[ns][3N], v (zeros), box[ns], dxdvdt[ns][3]"""
    P = np.asarray(P, dtype=np.float32)
    T = np.asarray(T, dtype=np.float32)
    nrows = len(P) - row0 if nrows is None else nrows
    nt = len(T)
    frac = fcc_fractional(sz)
    n = len(frac)
    ns = nrows * nt
    x = np.empty((ns, 3 * n))
    box = np.empty(ns)
    amp = dx * LAT[el][1]
    for r in range(nrows):
        i = row0 + r
        b = relax_box(sz, float(P[i]), el)
        for j in range(nt):
            k = r * nt + j
            rng = np.random.Generator(np.random.Philox(key=[seed, i * nt + j]))
            xx = frac * b + amp * 2.0 * (rng.random((n, 3)) - 0.5)
            xx -= np.floor(xx / b) * b
            bk = b
            if interpolate:
                # -is (remcmc:409-419): expand the volume by exp(0.75 (j+1)/NT) about the origin; the 1024-step NVE run with
                # fresh velocities that follows (remcmc:421-425) is Engine.run_md
                bk = float(np.cbrt(np.exp(np.log(b ** 3) + 0.75 * (j + 1) / nt)))
                xx = xx * (bk / b)
            x[k] = xx.reshape(-1)
            box[k] = bk
    v = np.zeros_like(x)
    d = np.tile(np.array([dx, dv, TIMESTEP[UNITS[el]]]), (ns, 1))
    return x, v, box, d
