"""GPU parity tests: the HIP path (through the C-ABI, libnm_hip.so) against the CPU oracle on the same seeded
inputs.  Tolerance: 1e-6 relative on energies / criteria (BASELINE.json north_star); decisions, counters,
branches and exchange permutations must be identical."""
import numpy as np
import pytest

from helpers import OracleLoop, grids

pytestmark = pytest.mark.gpu

RTOL = 1e-6  # north_star tolerance for floating point


def make_engine(loop, sz, P, T, **kw):
    import neuralmelting_amd as nm
    e = nm.Engine(4 * sz ** 3, P, T, row0=loop.row0, nrows=loop.nrows, seed=loop.seed, **kw)
    e.set_state(loop.x, loop.v, loop.box, loop.d)
    return e


def test_library_is_the_hip_one():
    import neuralmelting_amd._lib as B
    L = B.load()
    assert B.LIB_PATH.endswith('libnm_hip.so')
    for s in B.SYMBOLS:
        assert hasattr(L, s)


@pytest.mark.parametrize('sz', [4, 6, 8])
def test_eval_parity(oracle, sz):
    """batched lj_energy_force (a-1) vs the oracle's list evaluation and direct sum"""
    P, T = grids(2, 2)
    loop = OracleLoop(oracle, sz, P, T)
    e = make_engine(loop, sz, P, T)
    U, W, f = e.eval()
    n = loop.natoms
    for k in range(loop.ns):
        s = oracle.Sim(n)
        s.set_box(loop.box[k]); s.set_x(loop.x[k]); s.setup()
        assert abs(U[k] - s.pe) <= 1e-11 * abs(s.pe)
        assert abs(W[k] - s.virial) <= 1e-10 * abs(s.virial)
        fo = s.get_f()
        np.testing.assert_allclose(f[k], fo, rtol=0, atol=1e-10 * np.abs(fo).max())
    e.close()


def test_eval_known_answer_fcc():
    """SURVEY.md §8c known answers straight through the HIP path"""
    import neuralmelting_amd as nm
    from neuralmelting_amd import lattice
    P, T = grids(1, 1)
    for sz, pairs in ((4, 9984), (6, 33696), (8, 79872)):
        box = sz * lattice.lattice_constant('LJ')
        x = (lattice.fcc_fractional(sz) * box).reshape(1, -1)
        n = 4 * sz ** 3
        e = nm.Engine(n, P, T)
        e.set_state(x, np.zeros_like(x), [box], [[0.03125, 0.03125, 0.00390625]])
        U, W, f = e.eval()
        assert abs(U[0] / n - (-8.034879297835)) < 1e-10
        assert abs(W[0] / (3 * box ** 3) - 3.540508884) < 1e-8
        assert np.abs(f).max() < 1e-9
        st = e.stats()
        assert st[0, 3] == pairs
        e.close()


@pytest.mark.parametrize('sz', [4, 6])
def test_eval_lammps_bench_lj_step0(sz):
    """the step-0 thermo line of LAMMPS's own bench/in.lj logs (E_pair -6.7733681, Press -5.0197073; see tests/test_oracle.py for
    where the numbers come from and what they pin) through nm_eval"""
    import neuralmelting_amd as nm
    from test_oracle import LAMMPS_BENCH_LJ as B, bench_lj_lattice
    x, box = bench_lj_lattice(sz)
    n = len(x)
    P, T = grids(1, 1)
    e = nm.Engine(n, P, T)
    e.set_state(x.reshape(1, -1), np.zeros((1, 3 * n)), [box], [[0.03125, 0.03125, 0.00390625]])
    U, W, f = e.eval()
    e.close()
    assert abs(U[0] / n - B['e_pair']) < 5e-8
    assert abs(B['rho'] * B['T'] * (1.0 - 1.0 / B['N']) + W[0] / (3.0 * box ** 3) - B['press']) < 5e-8
    assert np.abs(f).max() < 1e-9


@pytest.mark.parametrize('bulk', [True, False])
def test_block_trace_parity(oracle, bulk):
    """one block of moves, move by move: branch, decision, criterion, energy after (a-2..a-7)"""
    sz, mod = 4, 24
    P, T = grids(2, 2)
    kw = dict(bulk=bulk, ppos=0.25, pvol=0.25)
    loop = OracleLoop(oracle, sz, P, T, **kw)
    e = make_engine(loop, sz, P, T, **kw)
    e.set_trace(True)
    e.set_step(0)
    e.run_block(mod)
    rows = e.thermo()
    tr = e.trace(mod)
    n = loop.natoms
    for k in range(loop.ns):
        s = oracle.Sim(n)
        s.set_rng(loop.seed, loop.row0 * loop.nt + k, 0)
        out = s.run_block(loop.x[k], loop.v[k], loop.box[k], loop.d[k], mod=mod, nstps=8, bulk=bulk, ppos=0.25, pvol=0.25,
                          lat=1.122, t=loop.tq[k], et=loop.et[k], pf=loop.pf[k], trace=True)
        to = out['trace']
        np.testing.assert_array_equal(tr[k, :, 0], to[:, 0])           # branch
        np.testing.assert_array_equal(tr[k, :, 1], to[:, 1])           # accepted
        np.testing.assert_allclose(tr[k, :, 2], to[:, 2], rtol=RTOL, atol=1e-7)
        np.testing.assert_allclose(tr[k, :, 3], to[:, 3], rtol=RTOL)
        np.testing.assert_allclose(rows[k, :5], out['thermo'], rtol=RTOL)
        np.testing.assert_array_equal(rows[k, 8:14], out['counters'])
        np.testing.assert_array_equal(rows[k, 14:17].astype(np.float32), out['ratios'])
    x, v, box, d = e.get_state()
    e.close()


@pytest.mark.parametrize('cus', [1, 2, 4, 8])
def test_iterative_moves_at_once_parity(oracle, monkeypatch, cus):
    """a-3 in reference mode: the device evaluates the N trials of a move at once (Replica::iter_pmc_all: the positions do not
    depend on the decisions because the reference never undoes a trial), shared out over the cluster at 2 / 4 / 8 workgroups per
    replica; the oracle runs them one after the other as the reference does.  Position-move-heavy blocks, two of them, so that
    image flags, counters and step sizes of the first feed the second; final positions and velocities compared as well."""
    monkeypatch.setenv('NM_CUS_PER_REPLICA', str(cus))
    sz, mod = 4, 12
    P, T = grids(2, 2)
    kw = dict(bulk=False, ppos=0.5, pvol=0.2)
    loop = OracleLoop(oracle, sz, P, T, **kw)
    e = make_engine(loop, sz, P, T, **kw)
    assert e.cus_per_replica == cus
    for step in range(2):
        e.set_step(step)
        e.run_block(mod)
        rows = e.thermo()
        loop.run_block(mod, step)
        ro = loop.rows()
        np.testing.assert_allclose(rows[:, :8], ro[:, :8], rtol=RTOL)
        np.testing.assert_array_equal(rows[:, 8:], ro[:, 8:])       # counters and float32 ratios
        e.adapt()
        loop.adapt()
    x, v, box, d = e.get_state()
    np.testing.assert_allclose(x, loop.x, rtol=0, atol=1e-8)
    np.testing.assert_allclose(v, loop.v, rtol=0, atol=1e-7)
    np.testing.assert_array_equal(d, loop.d)
    e.close()


@pytest.mark.parametrize('nstps', [1, 2, 3])
@pytest.mark.parametrize('cus', [1, 2, 8])
def test_short_trajectories_parity(oracle, monkeypatch, nstps, cus):
    """NSTPS = 1, 2, 3 (-ts): the first / last / only steps of a trajectory take different paths through the integrator
    hand-over (first kick outside the pair loop, middle steps fused into it, the last evaluation carrying the kinetic energy);
    each with one, two and eight workgroups per replica"""
    monkeypatch.setenv('NM_CUS_PER_REPLICA', str(cus))
    sz, mod = 4, 16
    P, T = grids(2, 2)
    kw = dict(bulk=True, ppos=0.1, pvol=0.1, nstps=nstps)
    loop = OracleLoop(oracle, sz, P, T, **kw)
    e = make_engine(loop, sz, P, T, **kw)
    assert e.cus_per_replica == cus
    e.set_trace(True)
    for step in range(2):
        e.set_step(step)
        e.run_block(mod)
        loop.run_block(mod, step)
        rows, ref = e.thermo(), loop.rows()
        np.testing.assert_array_equal(rows[:, 8:14], ref[:, 8:14])
        np.testing.assert_allclose(rows[:, :5], ref[:, :5], rtol=RTOL)
        x, v, box, d = e.get_state()
        np.testing.assert_allclose(x, loop.x, rtol=0, atol=1e-9)
        np.testing.assert_allclose(v, loop.v, rtol=0, atol=1e-8)
        e.adapt()
        loop.adapt()
    e.close()


def test_cycles_parity_with_exchange(oracle):
    """three full cycles of the main loop (remcmc:977-995): gen_samples, thermo rows, gen_mc_params, exchange"""
    sz, mod, ncyc = 4, 16, 3
    P, T = grids(2, 4)
    loop = OracleLoop(oracle, sz, P, T)
    e = make_engine(loop, sz, P, T)
    for step in range(ncyc):
        e.set_step(step)
        e.run_block(mod)
        rows = e.thermo()
        loop.run_block(mod, step)
        ro = loop.rows()
        np.testing.assert_allclose(rows[:, :8], ro[:, :8], rtol=RTOL)
        np.testing.assert_array_equal(rows[:, 8:], ro[:, 8:])
        e.adapt()
        loop.adapt()
        nsw = e.exchange()
        swaps, perm, crit = loop.exchange(step)
        assert nsw == swaps
        np.testing.assert_allclose(e.exchange_crit(), crit, rtol=RTOL, atol=1e-9)
    x, v, box, d = e.get_state()
    np.testing.assert_allclose(box, loop.box, rtol=1e-12)
    np.testing.assert_array_equal(d, loop.d)
    # configurations agree up to the accumulated round-off of different summation orders
    np.testing.assert_allclose(x, loop.x, rtol=0, atol=1e-7)
    e.close()


def test_tape_replay_parity(oracle):
    """externally supplied uniforms (the order of the reference's np.random calls) drive both paths identically"""
    sz, mod = 4, 20
    P, T = grids(1, 2)
    loop = OracleLoop(oracle, sz, P, T)
    rng = np.random.default_rng(11)
    tapes = []
    for k in range(loop.ns):
        tapes.append(rng.random(4 * mod))  # tags are stored as randint/65536, so every entry is a uniform
    e = make_engine(loop, sz, P, T)
    e.set_rng_tape(tapes)
    e.set_trace(True)
    e.run_block(mod)
    tr = e.trace(mod)
    rows = e.thermo()
    for k in range(loop.ns):
        s = oracle.Sim(loop.natoms)
        s.set_rng(loop.seed, k, 0)
        out = s.run_block(loop.x[k], loop.v[k], loop.box[k], loop.d[k], mod=mod, nstps=8, bulk=True, ppos=0.125, pvol=0.125,
                          lat=1.122, t=loop.tq[k], et=loop.et[k], pf=loop.pf[k], tape=tapes[k], trace=True)
        np.testing.assert_array_equal(tr[k, :, :2], out['trace'][:, :2])
        np.testing.assert_allclose(tr[k, :, 2:], out['trace'][:, 2:], rtol=RTOL, atol=1e-7)
        np.testing.assert_array_equal(rows[k, 8:14], out['counters'])
    e.close()


# every instantiation nm_api.hip's launch_kind can pick: 5^3 / 6^3 -> CfgMidH (Q = 1: half lists), CfgMid (Q = 2), CfgMidQ4, CfgMidQ8; 8^3 -> CfgLargeH (Q = 1: half lists), CfgLarge (Q = 2, 4)
LARGE_CELL_CASES = [(sz, q) for sz in (5, 6) for q in (1, 2, 4, 8)] + [(8, 1), (8, 2), (8, 4)]


@pytest.mark.parametrize('bulk', [True, False])
@pytest.mark.parametrize('sz,cus', LARGE_CELL_CASES)
def test_block_parity_large_cells(oracle, monkeypatch, sz, cus, bulk):
    """the HBM-list kernels (5^3: 500 atoms = the reference's run.sh size, a ragged fit for 64-lane waves; 6^3: 864 atoms;
    8^3: 2048 atoms) against the oracle, bulk and iterative position moves, at every workgroups-per-replica setting the size
    has an instantiation for"""
    monkeypatch.setenv('NM_CUS_PER_REPLICA', str(cus))
    mod = 6
    P, T = grids(1, 2)
    kw = dict(bulk=bulk, ppos=0.3, pvol=0.2)
    loop = OracleLoop(oracle, sz, P, T, **kw)
    e = make_engine(loop, sz, P, T, **kw)
    assert e.cus_per_replica == cus
    e.set_trace(True)
    e.run_block(mod)
    rows = e.thermo()
    tr = e.trace(mod)
    loop.run_block(mod, 0)
    ro = loop.rows()
    np.testing.assert_allclose(rows[:, :8], ro[:, :8], rtol=RTOL)
    np.testing.assert_array_equal(rows[:, 8:], ro[:, 8:])
    x, v, box, d = e.get_state()
    np.testing.assert_allclose(x, loop.x, rtol=0, atol=1e-8)
    np.testing.assert_allclose(v, loop.v, rtol=0, atol=1e-7)
    e.close()


@pytest.mark.parametrize('nstps', [1, 3])
@pytest.mark.parametrize('sz,cus', LARGE_CELL_CASES)
def test_short_trajectories_parity_large_cells(oracle, monkeypatch, sz, cus, nstps):
    """the integrator hand-over of the HBM-list kernels: HMC-heavy blocks with one- and three-step trajectories (first kick
    outside the pair loop, fused middle steps, last evaluation carrying the kinetic energy) at every workgroups-per-replica
    setting, two blocks so that the second starts from velocities and step sizes the first one left"""
    monkeypatch.setenv('NM_CUS_PER_REPLICA', str(cus))
    mod = 5
    P, T = grids(1, 2)
    kw = dict(bulk=True, ppos=0.1, pvol=0.1, nstps=nstps)
    loop = OracleLoop(oracle, sz, P, T, **kw)
    e = make_engine(loop, sz, P, T, **kw)
    assert e.cus_per_replica == cus
    for step in range(2):
        e.set_step(step)
        e.run_block(mod)
        loop.run_block(mod, step)
        rows, ref = e.thermo(), loop.rows()
        np.testing.assert_array_equal(rows[:, 8:14], ref[:, 8:14])
        np.testing.assert_allclose(rows[:, :5], ref[:, :5], rtol=RTOL)
        x, v, box, d = e.get_state()
        np.testing.assert_allclose(x, loop.x, rtol=0, atol=1e-8)
        np.testing.assert_allclose(v, loop.v, rtol=0, atol=1e-7)
        e.adapt()
        loop.adapt()
    e.close()


@pytest.mark.parametrize('sz,cus', [(4, 2), (4, 4), (4, 8)] + [(sz, q) for (sz, q) in LARGE_CELL_CASES if q > 1])
def test_block_that_ends_on_a_rejected_trajectory(oracle, monkeypatch, sz, cus):
    """A block whose LAST move is a Hamiltonian trajectory that is rejected, at every instantiation with more than one workgroup per
    replica (so that some workgroup owns atoms a0 != 0): restore() then hands the saved velocities back on the elementwise mapping
    (atom i on thread i mod BLOCK) and the block's closing kinetic-energy sum reads them on the integrator's mapping (atom a0 + t on
    thread t) — one of the three hand-offs between the two mappings that were unordered until round 3 (nm_kernels.h NM_FOR_OWN; seen
    once as an 8e-5 error in temp / ke).  Only trajectories are drawn (PPOS = PVOL = 0) and the time step is five times the default:
    the energy error of every one of them is 10-17 kT, so it is rejected — which the test asserts from the trace before it compares the closing temp, ke and the
    velocities with the oracle's."""
    monkeypatch.setenv('NM_CUS_PER_REPLICA', str(cus))
    mod = 3
    P, T = grids(1, 2)
    kw = dict(bulk=True, ppos=0.0, pvol=0.0)
    loop = OracleLoop(oracle, sz, P, T, **kw)
    loop.d[:, 2] = 0.02                           # dt: 5 x TIMESTEP (12 x makes the hot replica's trajectories explode to 1e30 and NaN)
    e = make_engine(loop, sz, P, T, **kw)
    assert e.cus_per_replica == cus
    e.set_trace(True)
    for step in range(2):                         # the second block starts from the restored velocities of the first
        e.set_step(step)
        e.run_block(mod)
        rows, tr = e.thermo(), e.trace(mod)
        loop.run_block(mod, step)
        ro = loop.rows()
        assert (tr[:, -1, 0] == 2.0).all() and (tr[:, -1, 1] == 0.0).all(), 'the last move must be a rejected trajectory'
        np.testing.assert_array_equal(rows[:, 8:14], ro[:, 8:14])
        assert (rows[:, 13] == 0).all()           # nah: nothing was accepted
        np.testing.assert_allclose(rows[:, :5], ro[:, :5], rtol=RTOL)
        x, v, box, d = e.get_state()
        np.testing.assert_allclose(x, loop.x, rtol=0, atol=1e-8)
        np.testing.assert_allclose(v, loop.v, rtol=0, atol=1e-7)
        e.adapt()                                 # (dt shrinks by 1/16: still too long to be accepted)
        loop.adapt()
    e.close()


def test_init_md_parity(oracle):
    """the -is dynamics of init_sample (remcmc:421-425): velocities at T, then NVE, against the oracle's primitives"""
    from helpers import OracleEngine
    from neuralmelting_amd import remcmc
    sz, nsteps = 4, 96
    P, T = grids(1, 3)
    loop = OracleLoop(oracle, sz, P, T)
    e = make_engine(loop, sz, P, T)
    e.set_step(7)
    e.run_md(nsteps)
    rows = e.thermo()
    x, v, box, d = e.get_state()
    run = remcmc.Run(['-ss', '4', '-pn', '1', '-tn', '3'])
    oe = OracleEngine(oracle, run)
    oe.loop = loop
    oe.set_step(7)
    oe.run_md(nsteps)
    np.testing.assert_allclose(rows[:, :5], loop.thermo, rtol=RTOL)
    np.testing.assert_allclose(x, loop.x, rtol=0, atol=1e-8)
    np.testing.assert_allclose(v, loop.v, rtol=0, atol=1e-7)
    assert (rows[:, 8:] == 0).all()                      # no counters are touched
    e.close()


def test_sharded_rows_equal_single_context(oracle):
    """rows [0,1) and [1,2) run in two contexts reproduce the single-context result bit for bit (RNG keyed by global slot)"""
    sz, mod = 4, 8
    P, T = grids(2, 2)
    import neuralmelting_amd as nm
    full = OracleLoop(oracle, sz, P, T)
    e = make_engine(full, sz, P, T)
    e.run_block(mod); e.adapt(); e.exchange(); e.set_step(1); e.run_block(mod)
    ref = e.thermo()
    e.close()
    parts = []
    for r in range(2):
        lp = OracleLoop(oracle, sz, P, T, row0=r, nrows=1)
        ee = make_engine(lp, sz, P, T)
        ee.run_block(mod); ee.adapt(); ee.exchange(); ee.set_step(1); ee.run_block(mod)
        parts.append(ee.thermo())
        ee.close()
    np.testing.assert_array_equal(np.concatenate(parts), ref)


def test_failure_is_loud():
    """a box below 2*rc is outside the minimum-image regime: the engine reports it instead of returning numbers"""
    import neuralmelting_amd as nm
    P, T = grids(1, 1)
    n = 256
    x = np.random.default_rng(0).random((1, 3 * n)) * 4.0
    e = nm.Engine(n, P, T)
    e.set_state(x, np.zeros_like(x), [4.0], [[0.03, 0.03, 0.004]])
    with pytest.raises(nm.NMError):
        e.eval()
    e.close()
