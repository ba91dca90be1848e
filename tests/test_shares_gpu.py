"""BASELINE configs at the size one GPU holds of them: C3's per-GPU share (LJ 6^3, 2 x 16 rows = 32 replicas of
864 atoms, 8 workgroups per replica), C5's (LJ 8^3, 4 x 32 = 128 replicas of 2048 atoms, 2 workgroups per replica) and C4
(Al EAM, 8 x 8 replicas of 256 atoms, 4 workgroups per replica).  All three fill the chip exactly (nslots x Q = 256 workgroups,
the edge the cluster hand-over depends on); C5's share also as a grid of TWICE the chip (NM_OVERSUBSCRIBE=1: 4 workgroups per replica,
two rounds of clusters with a census each, longest block first).  One block against the oracle on a subset of the slots — the oracle needs seconds
per slot at these sizes — and the invariants + bit-identical replay on all of them."""
import numpy as np
import pytest

from helpers import constants_lj, constants_metal
from neuralmelting_amd import lattice

pytestmark = pytest.mark.gpu
RTOL = 1e-6

# name, element, supercell, pressure rows on this GPU, rows of the whole grid, temperatures, workgroups per replica, MOD, slots
SHARES = [
    ('C3', 'LJ', 6, 2, 16, 16, 8, 12, (0, 5, 15, 16, 26, 31)),
    ('C5', 'LJ', 8, 4, 32, 32, 2, 8, (0, 31, 45, 77, 100, 127)),
    ('C5x2', 'LJ', 8, 4, 32, 32, 4, 8, (0, 31, 64, 127)),
    ('C4', 'Al', 4, 8, 8, 8, 4, 24, (0, 7, 9, 28, 36, 54, 63)),
]


@pytest.fixture(autouse=True)
def oversubscribe(request, monkeypatch):
    if 'C5x2' in request.node.name:
        monkeypatch.setenv('NM_OVERSUBSCRIBE', '1')


def run_share(el, sz, nrows, np_all, nt, mod, cycles=1):
    import neuralmelting_amd as nm
    P = np.linspace(1.0, 8.0, np_all, dtype=np.float32)
    T = (np.linspace(0.25, 2.5, nt, dtype=np.float32) if el == 'LJ' else np.linspace(256.0, 2560.0, nt, dtype=np.float32))
    x, v, box, d = lattice.init_states(sz, P, T, 0.03125, 0.03125, el=el, row0=0, nrows=nrows)
    e = nm.Engine(4 * sz ** 3, P, T, element=el, row0=0, nrows=nrows)
    e.set_state(x, v, box, d)
    rows, perms = [], []
    for step in range(cycles):
        e.set_step(step)
        e.run_block(mod)
        rows.append(e.thermo())
        e.adapt()
        e.exchange()
        perms.append(e.perm())
    out = dict(P=P, T=T, x0=x, v0=v, box0=box, d0=d, rows=np.array(rows), perms=np.array(perms), state=e.get_state(),
               q=e.cus_per_replica, ns=e.nslots)
    e.close()
    return out


@pytest.mark.parametrize('name,el,sz,nrows,np_all,nt,q,mod,slots', SHARES, ids=[s[0] for s in SHARES])
def test_share_one_block_against_the_oracle(oracle, name, el, sz, nrows, np_all, nt, q, mod, slots):
    r = run_share(el, sz, nrows, np_all, nt, mod)
    n = 4 * sz ** 3
    assert r['ns'] == nrows * nt and r['q'] == q and r['ns'] * r['q'] == (512 if name == 'C5x2' else 256)  # the whole chip, no CU to spare (or twice)
    rows = r['rows'][0]
    et, pf, tq = (constants_lj if el == 'LJ' else constants_metal)(r['P'], r['T'], 0, nrows)
    kw = dict(units=1, mass=lattice.MASS['Al'], pot=1) if el == 'Al' else {}
    x1, v1, box1, _ = r['state']
    inv = np.argsort(r['perms'][0])          # the exchange after the block moved buffers between slots: slot k's block result now
    for k in slots:                           # sits where perm says; thermo rows were read before the exchange
        s = oracle.Sim(n, **kw)
        s.set_rng(256, k, 0)
        out = s.run_block(r['x0'][k], r['v0'][k], r['box0'][k], r['d0'][k], mod=mod, nstps=8, bulk=True, ppos=0.125, pvol=0.125,
                          lat=lattice.LAT[el][1], t=tq[k], et=et[k], pf=pf[k])
        np.testing.assert_array_equal(rows[k, 8:14], out['counters'])
        np.testing.assert_array_equal(rows[k, 14:17].astype(np.float32), out['ratios'])
        np.testing.assert_allclose(rows[k, :5], out['thermo'], rtol=RTOL)
        kk = inv[k]                           # the slot that holds this configuration after the sweep
        assert abs(box1[kk] - out['box']) <= 1e-12 * out['box']
        np.testing.assert_allclose(x1[kk], out['x'], rtol=0, atol=1e-8 * lattice.LAT[el][1])
    # invariants on every slot
    cnt = rows[:, 8:14]
    np.testing.assert_array_equal(cnt[:, 0] + cnt[:, 2] + cnt[:, 4], mod)
    assert (cnt[:, 1] <= cnt[:, 0]).all() and (cnt[:, 3] <= cnt[:, 2]).all() and (cnt[:, 5] <= cnt[:, 4]).all()
    assert np.isfinite(rows).all() and (rows[:, 1] < 0).all() and (rows[:, 0] > 0).all()
    p = r['perms'][0]
    assert sorted(p) == list(range(r['ns'])) and (p // nt == np.arange(r['ns']) // nt).all()


@pytest.mark.parametrize('name,el,sz,nrows,np_all,nt,q,mod,slots', SHARES, ids=[s[0] for s in SHARES])
def test_share_replays_bit_for_bit(name, el, sz, nrows, np_all, nt, q, mod, slots):
    """two cycles (block, adapt, exchange) twice from the same seeds: identical bits in every thermo row, label and coordinate"""
    a = run_share(el, sz, nrows, np_all, nt, mod, cycles=2)
    b = run_share(el, sz, nrows, np_all, nt, mod, cycles=2)
    np.testing.assert_array_equal(a['rows'], b['rows'])
    np.testing.assert_array_equal(a['perms'], b['perms'])
    for u, w in zip(a['state'], b['state']):
        np.testing.assert_array_equal(u, w)
    steps = a['rows'][:, :, 5:8]
    f = steps[1] / steps[0]
    assert np.isin(np.round(f, 6), [0.9375, 1.0, 1.0625]).all()


def test_equilibrated_8cubed_lists_keep_clear_of_their_slots():
    """The 2048-atom kernel keeps 224 list slots per atom (lists in HBM; skin 0.6 => list radius 3.1).  The densest states of the
    BASELINE grids are those of the highest pressure row, P* = 8: its 32 temperatures — cold crystal to hot dense liquid — run
    for 30 cycles (adaptation, exchange: equilibrated chains, HMC accepting), and the longest row any rebuild produced must stay
    well below the slots there are (nm_stats_get columns 8, 9).  An overflow would stop the run (NM_ST_LIST_OVERFLOW)."""
    import neuralmelting_amd as nm
    P = np.linspace(1.0, 8.0, 32, dtype=np.float32)
    T = np.linspace(0.25, 2.5, 32, dtype=np.float32)
    x, v, box, d = lattice.init_states(8, P, T, 0.03125, 0.03125, row0=31, nrows=1)
    e = nm.Engine(2048, P, T, row0=31, nrows=1)
    e.set_state(x, v, box, d)
    for step in range(30):
        e.set_step(step)
        e.run_block(64)
        e.adapt()
        e.exchange(count=False)
    e.synchronize()
    st = e.stats()
    rows = e.thermo()
    e.close()
    slots = st[0, 9]
    assert slots == 224
    assert st[:, 1].sum() > 30 * 32                      # the lists were rebuilt many times
    assert 130 <= st[:, 8].max() <= 0.85 * slots, st[:, 8].max()


def test_equilibrated_6cubed_lists_keep_clear_of_their_slots():
    """The 864-atom kernels keep 192 list slots per atom since round 4 (160 before) and run with a skin of 0.55 (list radius 3.05).  The densest
    states of BASELINE config 3 are those of its two highest pressure rows: their 2 x 16 replicas — C3's share of a GPU, 8 workgroups per replica,
    the lists in LDS — run for 30 cycles (equilibrated chains), and the longest row any rebuild produced must stay well below the slots there are."""
    import neuralmelting_amd as nm
    P = np.linspace(1.0, 8.0, 16, dtype=np.float32)
    T = np.linspace(0.25, 2.5, 16, dtype=np.float32)
    x, v, box, d = lattice.init_states(6, P, T, 0.03125, 0.03125, row0=14, nrows=2)
    e = nm.Engine(864, P, T, row0=14, nrows=2)
    assert e.cus_per_replica == 8
    e.set_state(x, v, box, d)
    for step in range(30):
        e.set_step(step)
        e.run_block(64)
        e.adapt()
        e.exchange(count=False)
    e.synchronize()
    st = e.stats()
    e.close()
    assert st[0, 9] == 192
    assert st[:, 1].sum() > 30 * 32
    assert 120 <= st[:, 8].max() <= 0.85 * 192, st[:, 8].max()
