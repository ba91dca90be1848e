"""Oracle-free pins of the arithmetic the reference delegates to LAMMPS, measured on the HIP path through the C-ABI.

The reference holds no tests and LAMMPS is absent, so the behaviours SURVEY.md Appendix C lists as assumed (C1-C6) cannot be
pinned by reference output.  Each test below would fail if the HIP path deviated from the assumed behaviour, and none of them
touches oracle/:
  C1 lj/cut unshifted, no tail      test_parity_gpu.py::test_eval_known_answer_fcc (U/N, P, pair counts of the perfect crystal)
  C2 thermo_pe extensive            ideal-gas <V> and the NPT virial pressure below (an intensive U would change both criteria)
  C4 velocity create / zero angular test_velocity_create_semantics_with_image_flags
  C5 thermo_temp, thermo_press      test_velocity_create_semantics..., test_npt_virial_pressure_equals_imposed_pressure
  C6 fix nve = kick-drift-kick      test_hmc_energy_error_is_second_order_in_dt
  volume_mc's weight (remcmc:576)   test_dilute_gas_volume_follows_the_npt_ideal_gas_law
"""
import numpy as np
import pytest

from neuralmelting_amd import lattice
from neuralmelting_amd.exchange import philox4x32_10, u01

pytestmark = pytest.mark.gpu

N = 256
S_VEL_A, S_VEL_B = 5, 6          # csrc/nm_device.h stream ids


def c2_engine(**kw):
    import neuralmelting_amd as nm
    P = np.linspace(1.0, 8.0, 8, dtype=np.float32)
    T = np.linspace(0.25, 2.5, 8, dtype=np.float32)
    x, v, box, d = lattice.init_states(4, P, T, 0.03125, 0.03125)
    e = nm.Engine(N, P, T, **kw)
    return e, P, T, x, v, box, d


def test_dilute_gas_volume_follows_the_npt_ideal_gas_law():
    """volume_mc (remcmc:552-595) on a gas so dilute that U = 0: uniform steps in ln V accepted with
    exp(-[beta P dV - (N+1) ln(V'/V)]) sample p(V) ~ V^N exp(-beta P V), hence <V> = (N+1)/(beta P).
    The prefactor N instead of N+1, an energy counted per atom, or a criterion built on the rounded box would each shift <V>
    by far more than the tolerance."""
    import neuralmelting_amd as nm
    P = np.linspace(0.5e-5, 2.0e-5, 8, dtype=np.float32)
    T = np.linspace(1.0, 2.5, 8, dtype=np.float32)
    e = nm.Engine(N, P, T, ppos=0.0, pvol=1.0)                 # every move is a volume move
    et, pf = e.constants()
    ns = e.nslots
    box = np.cbrt((N + 1) / pf)                                 # start at the expected volume
    g = np.array([[i, j, k] for i in range(7) for j in range(7) for k in range(7)], dtype=np.float64)[:N]
    rng = np.random.default_rng(5)
    x = np.array([((g + 0.5 + 0.2 * (rng.random((N, 3)) - 0.5)) / 7.0 * b).ravel() for b in box])
    d = np.tile([0.03125, 0.12, 0.00390625], (ns, 1))          # fixed ln V step (no adaptation: a plain Markov chain)
    e.set_state(x, np.zeros_like(x), box, d)
    mod, burn, cycles = 8, 40, 700
    vol, pe = [], []
    for step in range(burn + cycles):
        e.set_step(step)
        e.run_block(mod)
        r = e.thermo()
        if step >= burn:
            vol.append(r[:, 4]); pe.append(r[:, 1])
    rows = e.thermo()
    e.close()
    vol, pe = np.array(vol), np.array(pe)
    assert np.abs(pe).mean() < 0.05 * et.min()                  # an ideal gas for the criterion's purposes
    acc = rows[:, 11] / rows[:, 10]
    assert (acc > 0.2).all() and (acc < 0.8).all()
    red = vol * pf[None, :] / (N + 1)                           # reduced volume: mean 1, variance 1/(N+1)
    nb = 20
    bm = red.reshape(nb, cycles // nb, ns).mean(1)              # block means absorb the residual autocorrelation
    se = bm.std(0, ddof=1) / np.sqrt(nb)
    z = (bm.mean(0) - 1.0) / se
    assert (np.abs(z) < 5.0).all(), np.sort(np.abs(z))[-4:]
    tot = red.mean()
    tot_se = np.sqrt((se ** 2).sum()) / ns
    assert tot_se < 6e-4
    assert abs(tot - 1.0) < 4.0 * tot_se + 1.5e-4, (tot, tot_se)  # 1.5e-4: second virial coefficient at these densities
    assert abs(tot - N / (N + 1.0)) > 4.0 * tot_se              # ... and the test does tell N+1 from N
    np.testing.assert_allclose(red.var(0, ddof=1).mean(), 1.0 / (N + 1), rtol=0.1)


def test_npt_virial_pressure_equals_imposed_pressure():
    """In the ensemble volume_mc samples, p(V, s) ~ V^N exp(-beta (U + P V)), integrating d/dV [V^N exp(-beta U)] exp(-beta P V)
    by parts gives  P = < N kT / V - dU/dV >  exactly.  For `pair_style lj/cut 2.5` WITHOUT shift (pair_modify defaults,
    SURVEY.md C1) U jumps by u(rc) = -0.0163 whenever a pair crosses the cutoff, so
        -dU/dV = W / 3V + u(rc) rc / 3V * sum_pairs delta(r - rc)        (W = sum r.f, what thermo_press carries)
    and the impulsive term is worth ~0.5 rho^2 in these units — several standard errors at every state point.  The engine's
    `virial` column is LAMMPS's thermo_press = ((3N-3) k T_kin + W) / 3V (remcmc:386); adding the kinetic difference and the
    impulsive term back, sample by sample from the same row and the same coordinates, gives an estimator whose mean must be
    the slot's pressure at all 64 state points (solid, liquid, dense gas).  This pins together: W, the pressure formula and its
    3N-3 degrees of freedom, the extensive U and the (N+1) weight in the VMC criterion, the unshifted cutoff, and — through
    the kinetic temperature — the exact-T rescale and the rotation removed without rescale.  Production move mix, adaptation
    and exchange on."""
    e, P, T, x, v, box, d = c2_engine()
    e.set_state(x, v, box, d)
    mod, burn, cycles = 64, 24, 96
    rc, dl = 2.5, 0.02                                          # shell half width: narrow, g(r) of the cold crystal is steep at rc
    urc = 4.0 * (rc ** -12 - rc ** -6)
    iu = np.triu_indices(N, 1)
    samples, shell = [], []
    for step in range(burn + cycles):
        e.set_step(step)
        e.run_block(mod)
        if step >= burn:
            samples.append(e.thermo())
            xs, _, bs, _ = e.get_state(velocities=False)
            cnt = []
            for k in range(64):
                q = xs[k].reshape(N, 3)
                dd = q[:, None, :] - q[None, :, :]
                dd -= bs[k] * np.rint(dd / bs[k])
                r2 = (dd * dd).sum(-1)[iu]
                cnt.append(((r2 > (rc - dl) ** 2) & (r2 < (rc + dl) ** 2)).sum())
            shell.append(cnt)
        e.adapt()
        e.exchange(count=False)
    e.close()
    r = np.array(samples)                                       # [cycle][slot][17]
    shell = np.array(shell, dtype=np.float64)
    tkin, press, vol = r[:, :, 0], r[:, :, 3], r[:, :, 4]
    Tj = np.tile(T.astype(np.float64), 8)[None, :]
    Pi = np.repeat(P.astype(np.float64), 8)
    pvir = press + (N * Tj - (N - 1.0) * tkin) / vol
    pest = pvir + urc * rc / (3.0 * vol) * shell / (2.0 * dl)
    nb = 8

    def zscores(a):
        bm = a.reshape(nb, cycles // nb, 64).mean(1)
        se = bm.std(0, ddof=1) / np.sqrt(nb)
        # a standard error from eight block means is itself noisy (a t statistic with 7 degrees of freedom exceeds 5 in one of 64
        # slots every tenth run): no slot is trusted to be more precise than the typical one
        se = np.maximum(se, np.median(se))
        return (bm.mean(0) - Pi) / se, bm.mean(0) - Pi
    z, diff = zscores(pest)
    assert (np.abs(z) < 5.0).all(), np.sort(np.abs(z))[-4:]    # measured: max 2.8-3.1 over the 64 slots
    assert abs(z.mean()) < 0.5, z.mean()                        # measured: -0.05 ... -0.09
    assert abs(diff.mean()) < 0.02, diff.mean()                 # measured: -0.0016 ... -0.0065 (pressures 1 ... 8)
    z0, diff0 = zscores(pvir)                                   # without the impulsive term the same data are off by ~0.4:
    assert z0.mean() > 4.0 and diff0.mean() > 0.25              # a shifted potential, or W off by a factor, cannot pass
    # kinetic temperature: velocities are drawn at exactly T with 3N-3 degrees of freedom and lose the rotation
    # (3 more, no rescale): <T_kin/T> = (3N-6)/(3N-3) = 0.99609 within sampling error (measured 0.99618 +- 0.0003);
    # 3N degrees of freedom in thermo_temp would read 0.9922, a rescale after the rotation removal 1.0000
    tr = (tkin / Tj).mean()
    tr_se = (tkin / Tj).std(ddof=1) / np.sqrt(tkin.size)
    assert tr_se < 5e-4 and abs(tr - (3.0 * N - 6.0) / (3.0 * N - 3.0)) < 4.0 * tr_se, (tr, tr_se)


def test_hmc_energy_error_is_second_order_in_dt():
    """fix nve (kick-drift-kick, dtf = dt/2 . ftm2v/m): over a trajectory of FIXED length the energy error of velocity Verlet
    scales as dt^2.  Same state, same velocity draw (Philox is keyed by seed, slot, move), trajectories of 0.016 time units as
    4 x 0.004, 8 x 0.002 and 16 x 0.001: the criterion hamiltonian_mc tests, dH = (U+K)'/kT - (U+K)/kT (remcmc:618-622), must
    drop 4x per halving.  A first-order scheme (full kick, drift) gives 2x, a wrong half-kick constant no convergence at all.
    Then acceptance -> 1 for dt -> 0 and collapses for an oversized step.
    The states are cold crystals at the density that puts the cutoff midway between the 5th and 6th neighbour shells
    (rho = 1.164): the unshifted potential jumps by u(rc) = -0.0163 whenever a pair crosses rc, which no integrator conserves
    and which swamps the dt^2 term in a liquid (measured there: dH independent of dt, ~0.2 per trajectory)."""
    import neuralmelting_amd as nm
    P = np.linspace(1.0, 8.0, 8, dtype=np.float32)
    T = np.linspace(0.05, 0.2, 8, dtype=np.float32)
    a = 2.5 / (0.5 * (np.sqrt(5.0) + np.sqrt(6.0))) * np.sqrt(2.0)     # fcc cell edge with rc between shells 5 and 6
    box = np.full(64, 4.0 * a)
    rng = np.random.default_rng(17)
    x = (lattice.fcc_fractional(4)[None] * box[0] + 0.04 * (rng.random((64, N, 3)) - 0.5)).reshape(64, -1)
    v = np.zeros_like(x)
    d = np.tile([0.03125, 0.03125, 0.00390625], (64, 1))
    dh = []
    for nstps, dt in ((4, 0.004), (8, 0.002), (16, 0.001)):
        e = nm.Engine(N, P, T, ppos=0.0, pvol=0.0, nstps=nstps)
        dd = d.copy(); dd[:, 2] = dt
        e.set_state(x, v, box, dd)
        e.set_trace(True)
        e.run_block(1)
        tr = e.trace(1)
        assert (tr[:, 0, 0] == 2.0).all()                       # the move was an HMC trajectory
        dh.append(tr[:, 0, 2])
        e.close()
    dh = np.array(dh)
    assert (np.abs(dh[2]) > 1e-9).all()                         # far above round-off
    r1, r2 = dh[0] / dh[1], dh[1] / dh[2]
    assert 3.8 < np.median(r1) < 4.2 and 3.9 < np.median(r2) < 4.1, (np.median(r1), np.median(r2))
    assert np.mean((r2 > 3.5) & (r2 < 4.5)) > 0.9, np.sort(r2)
    for dt, lo, hi in ((0.0005, 0.98, 1.0), (0.04, 0.0, 0.5)):
        e = nm.Engine(N, P, T, ppos=0.0, pvol=0.0, nstps=8)
        dd = d.copy(); dd[:, 2] = dt
        e.set_state(x, v, box, dd)
        e.run_block(32)
        r = e.thermo()
        e.close()
        assert (r[:, 12] == 32).all()
        acc = r[:, 13].sum() / r[:, 12].sum()
        assert lo <= acc <= hi, (dt, acc)


def lammps_velocity_create(xu, t, seed, gslot, tag, step, mass=1.0, kB=1.0, mvv2e=1.0):
    """velocity all create t SEED dist gaussian (mom yes, rot no, loop all) / zero linear / zero angular as LAMMPS's
    velocity.cpp documents them, in numpy, on UNWRAPPED coordinates xu[N][3]; the normal deviates are the engine's own
    per-atom Philox + Box-Muller streams (DESIGN.md §4 RNG)."""
    n = len(xu)
    v = np.empty((n, 3))
    for i in range(n):
        a = philox4x32_10((i, S_VEL_A, tag, step), (seed, gslot))
        b = philox4x32_10((i, S_VEL_B, tag, step), (seed, gslot))
        u1, u2 = u01(a[0], a[1]), u01(a[2], a[3])
        r = np.sqrt(-2.0 * np.log(1.0 - u1))
        v[i, 0], v[i, 1] = r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)
        u1, u2 = u01(b[0], b[1]), u01(b[2], b[3])
        v[i, 2] = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2 * np.pi * u2)
    v /= np.sqrt(mass)
    v -= v.mean(0)                                              # momentum yes
    tcur = mass * (v * v).sum() * mvv2e / ((3 * n - 3) * kB)
    v *= np.sqrt(t / tcur)                                      # scale to exactly t, dof = 3N - 3
    v -= v.mean(0)                                              # velocity all zero linear
    xc = xu - xu.mean(0)                                        # velocity all zero angular, about the centre of mass
    L = mass * np.cross(xc, v).sum(0)
    I = mass * ((xc * xc).sum() * np.eye(3) - xc.T @ xc)
    w = np.linalg.solve(I, L)
    return v - np.cross(w, xc)


def test_velocity_create_semantics_with_image_flags():
    """One bulk position move (accepted: tape uniform 0) carries atoms across the box faces, so their LAMMPS image flags
    are non-zero; the HMC move that follows draws velocities and integrates with timestep 0, so the block returns exactly what
    `velocity create / zero linear / zero angular` left.  They must (i) carry no linear momentum, (ii) carry no angular
    momentum about the centre of mass of the UNWRAPPED coordinates (with the wrapped ones they do), (iii) equal LAMMPS's
    documented sequence restated in numpy, (iv) sit at T_kin = T (1 - E_rot/K) < T with dof 3N-3, which the `temp`, `ke` and
    `virial` columns must reproduce from the returned x, v."""
    import neuralmelting_amd as nm
    P = np.linspace(1.0, 8.0, 2, dtype=np.float32)
    T = np.linspace(0.5, 2.0, 2, dtype=np.float32)
    x, v, box, d = lattice.init_states(4, P, T, 0.03125, 0.03125)
    box = np.round(box, 6)                                      # init_lammps hands the box over as '%f' (remcmc:466)
    x = np.mod(x.reshape(4, N, 3), box[:, None, None]).reshape(4, -1)
    d[:, 2] = 0.0                                               # timestep 0.000000
    tag_a, tag_b = 12345, 54321
    tape = [0.0, tag_a / 65536.0, 0.0, 0.99, tag_b / 65536.0, 0.5]   # roll, randint, accept | roll, randint, accept
    e = nm.Engine(N, P, T, seed=77)
    e.set_state(x, v, box, d)
    e.set_rng_tape([tape] * 4)
    e.set_trace(True)
    e.set_step(9)
    e.run_block(2)
    tr = e.trace(2)
    rows = e.thermo()
    xo, vo, bo, _ = e.get_state()
    e.close()
    np.testing.assert_array_equal(tr[:, :, 0], [[0.0, 2.0]] * 4)   # bulk PMC, then HMC
    np.testing.assert_array_equal(tr[:, :, 1], 1.0)                 # both accepted
    np.testing.assert_array_equal(bo, box)
    crossed = 0
    for k in range(4):
        L = box[k]
        xi, xw, vv = x[k].reshape(N, 3), xo[k].reshape(N, 3), vo[k].reshape(N, 3)
        dxy = xw - xi
        dxy -= L * np.rint(dxy / L)
        xu = xi + dxy                                           # unwrapped = wrapped + image * L
        crossed += int((np.abs(xu - xw) > 0.5 * L).any(1).sum())
        assert np.abs(vv.sum(0)).max() < 1e-11                  # (i)
        Lu = np.cross(xu - xu.mean(0), vv).sum(0)
        Lw = np.cross(xw - xw.mean(0), vv).sum(0)
        assert np.abs(Lu).max() < 1e-9, Lu                      # (ii)
        assert np.abs(Lw).max() > 1e-3, Lw
        ref = lammps_velocity_create(xu, float('%f' % T[k % 2]), 77, k, tag_b, 9)
        np.testing.assert_allclose(vv, ref, rtol=0, atol=1e-11) # (iii)
        ke = 0.5 * (vv * vv).sum()
        tk = 2.0 * ke / (3 * N - 3)
        t = float('%f' % T[k % 2])
        assert t * (1.0 - 24.0 / (3 * N - 3)) < tk < t          # (iv) a few kT/2 of rotation removed (3 of 765 on average), nothing rescaled
        np.testing.assert_allclose(rows[k, 0], tk, rtol=1e-12)
        np.testing.assert_allclose(rows[k, 2], ke, rtol=1e-12)
        # thermo_press = (dof k T_kin + W) / 3V with W = sum over pairs of r.f, recomputed here from the returned coordinates
        dd = xw[:, None, :] - xw[None, :, :]
        dd -= L * np.rint(dd / L)
        r2 = (dd * dd).sum(-1)[np.triu_indices(N, 1)]
        r6i = 1.0 / r2[r2 < 6.25] ** 3
        W = (r6i * (48.0 * r6i - 24.0)).sum()
        np.testing.assert_allclose(rows[k, 3], ((3 * N - 3) * tk + W) / (3.0 * L ** 3), rtol=1e-10)
        np.testing.assert_allclose(rows[k, 1], (r6i * (4.0 * r6i - 4.0)).sum(), rtol=1e-11)
    assert crossed >= 8                                         # the image flags really were in play
