"""GPU property tests at BASELINE's full headline size (C2: 8x8 grid of 256-atom replicas, MOD = 128) and of the cluster path."""
import os

import numpy as np
import pytest

from helpers import OracleLoop, grids

pytestmark = pytest.mark.gpu


def run_c2(cycles=3, mod=128, **env):
    import neuralmelting_amd as nm
    from neuralmelting_amd import lattice
    P, T = grids(8, 8)
    x, v, box, d = lattice.init_states(4, P, T, 0.03125, 0.03125)
    e = nm.Engine(256, P, T)
    e.set_state(x, v, box, d)
    rows, perms = [], []
    for step in range(cycles):
        e.set_step(step)
        e.run_block(mod)
        rows.append(e.thermo())
        e.adapt()
        e.exchange()
        perms.append(e.perm())
    st = e.get_state()
    q = e.cus_per_replica
    e.close()
    return np.array(rows), np.array(perms), st, q


def test_c2_full_size_invariants_and_determinism():
    rows, perms, (x, v, box, d), q = run_c2()
    assert q == 4                                                   # 64 replicas x 4 workgroups = all 256 CUs
    cnt = rows[:, :, 8:14]
    np.testing.assert_array_equal(cnt[:, :, 0] + cnt[:, :, 2] + cnt[:, :, 4], 128)      # every move is a PMC, VMC or HMC trial
    assert (cnt[:, :, 1] <= cnt[:, :, 0]).all() and (cnt[:, :, 3] <= cnt[:, :, 2]).all() and (cnt[:, :, 5] <= cnt[:, :, 4]).all()
    ratios = rows[:, :, 14:17]
    np.testing.assert_array_equal(ratios[:, :, 2].astype(np.float32),
                                  (cnt[:, :, 5].astype(np.float32) / np.maximum(cnt[:, :, 4], 1).astype(np.float32)))
    assert np.isfinite(rows).all() and (rows[:, :, 1] < 0).all() and (rows[:, :, 0] > 0).all()
    assert (box >= 5.0).all()                                       # minimum-image regime
    np.testing.assert_allclose(rows[:, :, 4], (rows[:, :, 4] ** (1 / 3)) ** 3)
    # the sweep never leaves a pressure row and is a permutation inside it (remcmc:782-798)
    for p in perms:
        assert sorted(p) == list(range(64))
        assert (p // 8 == np.arange(64) // 8).all()
    # adaptive steps moved by exactly one factor per cycle (remcmc:726-745)
    steps = rows[:, :, 5:8]
    f = steps[1] / steps[0]
    assert np.isin(np.round(f, 6), [0.9375, 1.0, 1.0625]).all()
    # same seed, same machine, same binary: bit-identical replay (fixed summation orders everywhere)
    rows2, perms2, (x2, v2, box2, d2), _ = run_c2()
    np.testing.assert_array_equal(rows, rows2)
    np.testing.assert_array_equal(perms, perms2)
    np.testing.assert_array_equal(x, x2)
    np.testing.assert_array_equal(v, v2)


def test_velocities_have_no_net_momentum_and_target_temperature():
    """after an accepted or rejected HMC move the stored velocities come from `velocity create`: zero COM momentum"""
    rows, perms, (x, v, box, d), q = run_c2(cycles=1, mod=32)
    vv = v.reshape(64, 256, 3)
    assert np.abs(vv.sum(1)).max() < 1e-9
    temp = rows[0, :, 0]
    T = np.tile(np.linspace(0.25, 2.5, 8, dtype=np.float32), 8)
    assert (np.abs(temp / T - 1) < 0.35).all()                      # kinetic temperature near the slot's temperature


def test_cluster_matches_single_workgroup(oracle, monkeypatch):
    """4 CUs per replica vs 1 CU per replica: same chains (only the energy summation order differs)"""
    import neuralmelting_amd as nm
    P, T = grids(2, 4)
    outs = []
    for q in ('4', '1'):
        monkeypatch.setenv('NM_CUS_PER_REPLICA', q)
        loop = OracleLoop(oracle, 4, P, T)
        e = nm.Engine(256, P, T)
        assert e.cus_per_replica == int(q)
        e.set_state(loop.x, loop.v, loop.box, loop.d)
        e.set_trace(True)
        for step in range(2):
            e.set_step(step); e.run_block(48); e.adapt(); e.exchange()
        outs.append((e.thermo(), e.trace(48), e.get_state(), e.perm()))
        e.close()
    (r4, t4, s4, p4), (r1, t1, s1, p1) = outs
    np.testing.assert_array_equal(t4[:, :, :2], t1[:, :, :2])       # branches and decisions
    np.testing.assert_allclose(t4[:, :, 2:], t1[:, :, 2:], rtol=1e-9, atol=1e-9)
    np.testing.assert_array_equal(p4, p1)
    np.testing.assert_allclose(s4[0], s1[0], rtol=0, atol=1e-9)
    np.testing.assert_array_equal(s4[3], s1[3])


def _dense_engine(np_rows, nt, box_edge, seed=3):
    """replicas of 256 atoms placed at random in a box so small that some atoms have more neighbours than list slots"""
    import neuralmelting_amd as nm
    P, T = grids(np_rows, nt)
    rng = np.random.default_rng(seed)
    ns = np_rows * nt
    x = rng.random((ns, 768)) * box_edge
    e = nm.Engine(256, P, T)
    e.set_state(x, np.zeros((ns, 768)), np.full(ns, box_edge), np.tile([0.03125, 0.03125, 0.00390625], (ns, 1)))
    return nm, e


@pytest.mark.parametrize('np_rows,nt', [(8, 8), (16, 8), (32, 8)])        # 4, 2 and 1 workgroups per replica (192 list slots)
def test_list_overflow_is_reported_by_the_whole_cluster_without_timeouts(np_rows, nt):
    """error path of the cluster hand-over: the workgroup whose list overflows poisons what it publishes, its peers take the
    status over and the replica leaves the block at once — NM_ERR_STATE with the reason, not a 2 s hand-over timeout"""
    import time
    nm, e = _dense_engine(np_rows, nt, 5.2)      # density 1.82: ~170 listed neighbours on average, tails beyond the 192 / 256 slots
    t0 = time.perf_counter()
    with pytest.raises(nm.NMError) as err:
        e.eval()
    assert err.value.code == -3 and 'neighbour list overflow' in str(err.value) and 'timed out' not in str(err.value)
    e.set_step(0)
    e.run_block(8)
    with pytest.raises(nm.NMError) as err:
        e.synchronize()
    dt = time.perf_counter() - t0
    assert err.value.code == -3 and 'neighbour list overflow' in str(err.value) and 'timed out' not in str(err.value)
    assert dt < 1.5, dt
    e.close()


def test_grid_that_cannot_be_resident_falls_back_to_fewer_workgroups_per_replica(monkeypatch):
    """Cluster launch safety.  NM_ASSUME_CUS makes nm_create believe the chip has 512 CUs, so it first tries 8 workgroups per
    replica for 64 replicas = 512 workgroups of 82 KB LDS each, twice what the chip can hold at once.  The residency probe
    (a census launch) must see that the grid does not gather, fall back to 4 per replica, say so in nm_create_note, and the
    chains must then be the ones a plain 4-per-replica context produces — no 2 s hand-over timeouts, no hang."""
    import time
    import neuralmelting_amd as nm
    from neuralmelting_amd import lattice
    P, T = grids(8, 8)
    x, v, box, d = lattice.init_states(4, P, T, 0.03125, 0.03125)
    outs = []
    for assume in ('512', None):
        if assume:
            monkeypatch.setenv('NM_ASSUME_CUS', assume)
        else:
            monkeypatch.delenv('NM_ASSUME_CUS')
        t0 = time.perf_counter()
        e = nm.Engine(256, P, T)
        note = e.note()
        assert e.lib.nm_last_error(e.h).decode() == ''  # a note is not an error
        assert e.cus_per_replica == 4
        if assume:
            assert '8 workgroups per replica (512 in all) did not gather' in note
            assert time.perf_counter() - t0 < 1.0
        else:
            assert note == ''
        e.set_state(x, v, box, d)
        e.run_block(16)
        outs.append(e.thermo())
        e.close()
    np.testing.assert_array_equal(outs[0], outs[1])


def test_a_block_that_ends_on_an_error_leaves_the_state_as_it_was(monkeypatch):
    """whatever stops a block (here an injected list overflow at the fourth rebuild, i.e. in the middle of it): x, v, box of the
    slots that stopped are those of the block's start, so a caller can inspect or re-issue; slots that finished have moved on"""
    import ctypes as C
    import neuralmelting_amd as nm
    from neuralmelting_amd import lattice
    P, T = grids(8, 8)
    x, v, box, d = lattice.init_states(4, P, T, 0.03125, 0.03125)
    e = nm.Engine(256, P, T)
    e.set_state(x, v, box, d)
    e.run_block(8)
    e.synchronize()
    x1, v1, box1, d1 = e.get_state()
    monkeypatch.setenv('NM_INJECT_OVERFLOW', '3,1')
    e.set_step(1)
    e.run_block(48)
    with pytest.raises(nm.NMError):
        e.synchronize()
    monkeypatch.delenv('NM_INJECT_OVERFLOW')
    st = e.status()
    stopped = st != 0
    assert 8 <= stopped.sum() < 64 and (st[stopped] == 1).all()            # NM_ST_LIST_OVERFLOW; the cold slots rebuild too rarely to be hit
    xo, vo, bo = np.empty((64, 768)), np.empty((64, 768)), np.empty(64)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    rc = e.lib.nm_get_state(e.h, 0, 64, dp(xo), dp(vo), dp(bo), None)     # the error was reported once; the context stays usable
    assert rc == 0
    np.testing.assert_array_equal(xo[stopped], x1[stopped])
    np.testing.assert_array_equal(vo[stopped], v1[stopped])
    np.testing.assert_array_equal(bo[stopped], box1[stopped])
    assert (np.abs(xo[~stopped] - x1[~stopped]).max(1) > 0).all()          # the others completed their 48 moves
    # the caller re-issues (the injection is off now): the block runs, nm_get_status is that of the LAST block, nothing is sticky
    e.set_step(2)
    e.run_block(8)
    e.synchronize()
    assert (e.status() == 0).all()
    rows = e.thermo()
    moves = rows[:, 8] + rows[:, 10] + rows[:, 12]         # (no nm_adapt in between: the counters run on)
    assert np.isfinite(rows).all() and (moves[stopped] == 8 + 8).all() and (moves[~stopped] == 8 + 48 + 8).all()
    e.close()


def test_nothing_runs_behind_a_block_that_stopped(monkeypatch):
    """a block stops on an error (injected overflow) with adapt, exchange and two more cycles already queued behind it: none of
    them may run on the stale state — the step sizes, the slot->buffer map and the configurations of ALL slots are those of the
    failed block's start for the slots that stopped, and exactly one block old for the others"""
    import neuralmelting_amd as nm
    from neuralmelting_amd import lattice
    P, T = grids(8, 8)
    x, v, box, d = lattice.init_states(4, P, T, 0.03125, 0.03125)
    e = nm.Engine(256, P, T)
    e.set_state(x, v, box, d)
    e.run_block(8)
    e.adapt()
    e.exchange(count=False)
    e.synchronize()
    x1, v1, box1, d1 = e.get_state()
    perm1 = e.perm()
    monkeypatch.setenv('NM_INJECT_OVERFLOW', '3,1')
    for step in (1, 2, 3):
        e.set_step(step)
        e.run_block(48)
        e.adapt()
        e.exchange(count=False)
    with pytest.raises(nm.NMError):
        e.synchronize()
    monkeypatch.delenv('NM_INJECT_OVERFLOW')
    stopped = e.status() != 0
    assert stopped.any() and not stopped.all()
    x2, v2, box2, d2 = e.get_state()
    np.testing.assert_array_equal(e.perm(), perm1)             # no exchange ran
    np.testing.assert_array_equal(d2, d1)                      # no adapt ran
    np.testing.assert_array_equal(x2[stopped], x1[stopped])
    # the replicas that completed the failed block ran it ONCE (48 moves), not three times: same as a clean context does
    f = nm.Engine(256, P, T)
    f.set_state(x1, v1, box1, d1)
    f.set_step(1)
    f.run_block(48)
    x3 = f.get_state()[0]
    f.close()
    np.testing.assert_array_equal(x2[~stopped], x3[~stopped])
    # the launches queued behind the failed block left before their residency census: the next launch must not mistake the short count for a grid
    # that is not resident (it would be re-issued at fewer workgroups per replica)
    e.set_step(4)
    e.run_block(8)
    e.synchronize()
    assert (e.status() == 0).all() and e.lib.nm_cus_per_replica(e.h) == 4 and e.heals == 0 and e.note() == ''
    e.close()


def test_a_launch_that_is_not_resident_is_reissued_with_fewer_workgroups(monkeypatch):
    """Self-healing cluster launches.  The residency census of the context's second cluster launch is made to fail
    (NM_INJECT_CENSUS=1: what a CU mask or another process taking CUs between nm_create and the launch does) with the next
    two cycles already queued behind it.  Nothing has been touched and nothing behind it runs; the next host call re-creates the
    launch at 2 workgroups per replica, re-issues the block and everything behind it, says so in nm_create_note — and the
    chains are bit for bit those of a context that ran the first cycle at 4 per replica and the rest at 2."""
    import neuralmelting_amd as nm
    from neuralmelting_amd import lattice
    P, T = grids(8, 8)
    x, v, box, d = lattice.init_states(4, P, T, 0.03125, 0.03125)
    mod = 12

    def cycle(e, step):
        e.set_step(step)
        e.run_block(mod)
        e.adapt()
        e.exchange(count=False)

    monkeypatch.setenv('NM_INJECT_CENSUS', '1')
    a = nm.Engine(256, P, T)
    assert a.cus_per_replica == 4 and a.note() == ''
    a.set_state(x, v, box, d)
    for step in range(3):
        cycle(a, step)
    a.synchronize()                                            # notices, re-issues cycles 1 and 2 at Q = 2
    monkeypatch.delenv('NM_INJECT_CENSUS')
    assert a.cus_per_replica == 2
    assert 'step 1 stopped at 4 workgroups per replica (grid not resident); re-issued at 2' in a.note()
    assert a.lib.nm_last_error(a.h).decode() == ''
    assert (a.status() == 0).all()
    ra, (xa, va, ba, da) = a.thermo(), a.get_state()
    a.close()

    b = nm.Engine(256, P, T)                                   # cycle 0 at 4 workgroups per replica
    b.set_state(x, v, box, d)
    cycle(b, 0)
    rb, (xb, vb, bb, db) = b.thermo(), b.get_state()
    b.close()
    monkeypatch.setenv('NM_CUS_PER_REPLICA', '2')
    c = nm.Engine(256, P, T)                                   # cycles 1, 2 at 2
    assert c.cus_per_replica == 2
    c.set_state(xb, vb, bb, db)
    c.set_thermo(rb[:, :5])
    cycle(c, 1)
    cycle(c, 2)
    rc_, (xc, vc, bc, dc) = c.thermo(), c.get_state()
    c.close()
    np.testing.assert_array_equal(xa, xc)
    np.testing.assert_array_equal(va, vc)
    np.testing.assert_array_equal(ba, bc)
    np.testing.assert_array_equal(da, dc)
    np.testing.assert_array_equal(ra[:, :8], rc_[:, :8])


@pytest.mark.parametrize('first', ['perm', 'trace', 'stats', 'status', 'counters', 'slots'])
def test_every_getter_and_setter_looks_at_a_halted_queue_first(monkeypatch, first):
    """include/nm.h promises that the next nm_synchronize / nm_get_* / nm_set_* after a halted block re-issues it.  Here the FIRST host call
    after the failed launch (census injection, two more cycles queued behind it) is one of the entry points that used to synchronise the
    stream only: it must see the healed queue's data — the permutation, trace and statistics of a context that was never disturbed —
    and a setter must not get in before the replay.  Timing counts the launches that did the work, nm_heal_count says there was a heal."""
    import neuralmelting_amd as nm
    from neuralmelting_amd import lattice
    P, T = grids(8, 8)
    x, v, box, d = lattice.init_states(4, P, T, 0.03125, 0.03125)
    mod = 6

    def cycles(e):
        e.set_trace(True)
        for step in range(3):
            e.set_step(step)
            e.run_block(mod)
            e.adapt()
            e.exchange(count=False)

    monkeypatch.setenv('NM_CUS_PER_REPLICA', '2')
    ref = nm.Engine(256, P, T)                                  # never disturbed, at the Q the healed context ends up with
    ref.set_state(x, v, box, d)
    ref.timing_reset()
    cycles(ref)
    want = dict(perm=ref.perm(), trace=ref.trace(mod), stats=ref.stats(), thermo=ref.thermo(), timing=ref.timing())
    ref.close()
    monkeypatch.delenv('NM_CUS_PER_REPLICA')

    monkeypatch.setenv('NM_INJECT_CENSUS', '0')                 # the context's FIRST cluster launch fails its census
    a = nm.Engine(256, P, T)
    assert a.cus_per_replica == 4 and a.heals == 0
    a.set_state(x, v, box, d)
    a.timing_reset()
    cycles(a)
    monkeypatch.delenv('NM_INJECT_CENSUS')
    if first == 'perm':
        np.testing.assert_array_equal(a.perm(), want['perm'])
    elif first == 'trace':
        np.testing.assert_array_equal(a.trace(mod), want['trace'])
    elif first == 'stats':
        np.testing.assert_array_equal(a.stats()[:, :4], want['stats'][:, :4])
    elif first == 'status':
        assert (a.status() == 0).all()
    elif first == 'counters':                                   # a setter: must land AFTER the replayed blocks, not under them
        a.set_counters(count=np.full((64, 6), 3.0))
        assert (a.thermo()[:, 8:14] == 3.0).all()
    else:
        xs = a.get_slots([5, 9])[0]
        assert np.isfinite(xs).all()
    assert a.heals == 1 and a.cus_per_replica == 2
    np.testing.assert_array_equal(a.perm(), want['perm'])
    np.testing.assert_array_equal(a.trace(mod), want['trace'])
    if first != 'counters':
        np.testing.assert_array_equal(a.thermo(), want['thermo'])
    assert a.timing()[0] == want['timing'][0] == 3              # three launches did the work; the halted one and its no-op followers are not counted
    a.close()


def test_output_snapshots_equal_the_synchronous_reads_and_do_not_stop_the_stream():
    """nm_snapshot / nm_snapshot_fetch (write_outputs without stopping the stream): a snapshot queued right behind a block, fetched a whole cycle
    later — after nm_adapt has zeroed the counters, the exchange has re-labelled the slots and the next block has moved every atom — must hold
    exactly what nm_get_thermo / nm_get_state returned at that point of an identical run; at most two may be pending, and fetching without one is
    an error."""
    import neuralmelting_amd as nm
    from neuralmelting_amd import lattice
    P, T = grids(2, 4)
    x, v, box, d = lattice.init_states(4, P, T, 0.03125, 0.03125)

    def engine():
        e = nm.Engine(256, P, T)
        e.set_state(x, v, box, d)
        return e

    ref = engine()
    want = []
    for step in range(3):
        ref.set_step(step)
        ref.run_block(12)
        xs, _, bs, _ = ref.get_state(velocities=False)
        want.append((ref.thermo(), xs, bs))
        ref.adapt()
        ref.exchange(count=False)
    ref.close()

    e = engine()
    with pytest.raises(nm.NMError):
        e.snapshot_fetch()                                   # nothing pending
    got = []
    for step in range(3):
        e.set_step(step)
        e.run_block(12)
        e.snapshot()
        e.adapt()
        e.exchange(count=False)
        if step == 1:
            with pytest.raises(nm.NMError):
                e.snapshot()                                 # a third one while two are pending
            e.lib.nm_synchronize(e.h)                        # (clears the error text; the queue itself is fine)
        if step >= 1:
            got.append(e.snapshot_fetch())                   # the cycle before this one
    got.append(e.snapshot_fetch())
    e.close()
    for (r, xs, bs), (r0, x0, b0) in zip(got, want):
        np.testing.assert_array_equal(r, r0)
        np.testing.assert_array_equal(xs, x0)
        np.testing.assert_array_equal(bs, b0)


def test_census_counters_are_zeroed_before_they_could_wrap(monkeypatch):
    """The residency census counts on counters that only grow (no memset in front of a launch, round 4); the host zeroes them, and the
    bases the kernels compare with, once a base passes 0x3F000000 — 16 million launches into a run.  Here the counters start 300 below
    that value (NM_TEST_CENSUS_BASE, honoured under NM_TESTING only), so the second launch crosses it: every launch must still gather
    (no status, no heal) and the chains must be bit for bit those of a context that started from zero."""
    import neuralmelting_amd as nm
    from neuralmelting_amd import lattice
    P, T = grids(8, 8)
    x, v, box, d = lattice.init_states(4, P, T, 0.03125, 0.03125)

    def run():
        e = nm.Engine(256, P, T)
        assert e.cus_per_replica == 4
        e.set_state(x, v, box, d)
        for step in range(4):
            e.set_step(step)
            e.run_block(6)
            e.adapt()
            e.exchange(count=False)
        e.synchronize()
        out = (e.thermo(), e.get_state()[0], e.perm(), e.heals, e.note(), e.status().copy())
        e.close()
        return out

    ref = run()
    monkeypatch.setenv('NM_TEST_CENSUS_BASE', str(0x3F000000 - 300))
    got = run()
    assert got[3] == 0 and got[4] == '' and (got[5] == 0).all()
    np.testing.assert_array_equal(got[0], ref[0])
    np.testing.assert_array_equal(got[1], ref[1])
    np.testing.assert_array_equal(got[2], ref[2])


def test_a_grid_of_twice_the_chip_heals_to_the_resident_one(monkeypatch):
    """NM_OVERSUBSCRIBE=1: 128 replicas of 2048 atoms at 4 workgroups each run as two rounds of clusters, every cluster with its own
    residency census.  A launch whose censuses fail (injected) leaves the replicas untouched; the library re-issues it on the resident
    grid of 2 workgroups per replica, and the result is that of a context that ran there all along."""
    import neuralmelting_amd as nm
    from neuralmelting_amd import lattice
    P = np.linspace(1.0, 8.0, 32, dtype=np.float32); T = np.linspace(0.25, 2.5, 32, dtype=np.float32)
    x, v, box, d = lattice.init_states(8, P, T, 0.03125, 0.03125, row0=0, nrows=4)
    mod = 4

    def cycle(e, step):
        e.set_step(step)
        e.run_block(mod)
        e.adapt()
        e.exchange(count=False)

    monkeypatch.setenv('NM_OVERSUBSCRIBE', '1')
    monkeypatch.setenv('NM_INJECT_CENSUS', '1')
    a = nm.Engine(2048, P, T, row0=0, nrows=4)
    assert a.cus_per_replica == 4 and a.nslots == 128 and a.note() == ''
    a.set_state(x, v, box, d)
    for step in range(2):
        cycle(a, step)
    a.synchronize()                                            # notices, re-issues cycle 1 at Q = 2
    monkeypatch.delenv('NM_INJECT_CENSUS')
    assert a.cus_per_replica == 2
    assert 'step 1 stopped at 4 workgroups per replica (grid not resident); re-issued at 2' in a.note()
    assert (a.status() == 0).all()
    ra, (xa, va, ba, da) = a.thermo(), a.get_state()
    a.close()

    b = nm.Engine(2048, P, T, row0=0, nrows=4)                 # cycle 0 on the oversubscribed grid
    assert b.cus_per_replica == 4
    b.set_state(x, v, box, d)
    cycle(b, 0)
    rb, (xb, vb, bb, db) = b.thermo(), b.get_state()
    b.close()
    monkeypatch.delenv('NM_OVERSUBSCRIBE')
    c = nm.Engine(2048, P, T, row0=0, nrows=4)                 # cycle 1 on the resident one
    assert c.cus_per_replica == 2
    c.set_state(xb, vb, bb, db)
    c.set_thermo(rb[:, :5])
    cycle(c, 1)
    rc_, (xc, vc, bc, dc) = c.thermo(), c.get_state()
    c.close()
    np.testing.assert_array_equal(xa, xc)
    np.testing.assert_array_equal(va, vc)
    np.testing.assert_array_equal(ba, bc)
    np.testing.assert_array_equal(ra[:, :8], rc_[:, :8])


def test_box_smaller_than_twice_the_cutoff_is_refused():
    nm, e = _dense_engine(8, 8, 4.9)
    with pytest.raises(nm.NMError) as err:
        e.eval()
    assert err.value.code == -3 and 'box edge < 2*rc' in str(err.value)
    e.close()


@pytest.mark.parametrize('el,nth,q', [('LJ', n, q) for n in range(0, 12) for q in (0, 2)] + [('Al', n, 1) for n in (0, 1, 2, 5, 9)])
def test_list_overflow_anywhere_in_a_block_stops_the_whole_cluster(monkeypatch, el, nth, q):
    """Physical states of these systems do not overflow a list, so the error path is driven by fault injection
    (NM_INJECT_OVERFLOW=n,q: the n-th rebuild of a block "overflows" in workgroup q of every cluster).  Depending on n the
    rebuild falls into the first evaluation, a PMC/VMC energy evaluation or a force-only evaluation in the middle of an HMC
    trajectory, where that workgroup has nothing to publish but poison (publish_poison) — or, for the EAM, between its two
    passes.  In every case all workgroups of the cluster must stop at once: NM_ERR_STATE with the reason, no 2 s hand-over
    timeout, no hang."""
    import time
    import neuralmelting_amd as nm
    from neuralmelting_amd import lattice
    monkeypatch.setenv('NM_INJECT_OVERFLOW', '%d,%d' % (nth, q))
    P = np.linspace(1.0, 8.0, 8, dtype=np.float32)
    T = np.linspace(0.25, 2.5, 8, dtype=np.float32) if el == 'LJ' else np.linspace(256.0, 2560.0, 8, dtype=np.float32)
    x, v, box, d = lattice.init_states(4, P, T, 0.03125, 0.03125, el=el)
    e = nm.Engine(256, P, T, element=el)
    assert e.cus_per_replica == 4
    e.set_state(x, v, box, d)
    t0 = time.perf_counter()
    e.set_step(0)
    e.run_block(24)
    with pytest.raises(nm.NMError) as err:
        e.synchronize()
    dt = time.perf_counter() - t0
    assert err.value.code == -3 and 'neighbour list overflow' in str(err.value) and 'timed out' not in str(err.value)
    assert dt < 1.0, dt
    monkeypatch.delenv('NM_INJECT_OVERFLOW')
    e.close()


def test_long_run_statistics_agree_with_the_oracle(oracle):
    """Beyond the horizon where the two chains are still identical move by move (rounding differences are amplified by the
    dynamics after a few cycles) they must remain two samples of the SAME ensemble: 8x8 grid, 28 cycles of 64 moves with
    adaptation and replica exchange on both sides, same seeds; per-slot means of U and V over the last 20 cycles compared in
    units of their own standard error, acceptance ratios and adapted step sizes compared directly."""
    import neuralmelting_amd as nm
    P, T = grids(8, 8)
    cycles, mod, burn = 28, 64, 8
    lp = OracleLoop(oracle, 4, P, T)
    e = nm.Engine(256, P, T)
    e.set_state(lp.x, lp.v, lp.box, lp.d)
    g_rows, o_rows, same = [], [], 0
    for step in range(cycles):
        e.set_step(step)
        e.run_block(mod)
        lp.run_block(mod, step)
        g, o = e.thermo(), lp.rows()
        if np.array_equal(g[:, 8:14], o[:, 8:14]) and same == step:
            same += 1                                       # cycles in which every counter of every replica still agrees
        g_rows.append(g)
        o_rows.append(o)
        e.adapt()
        lp.adapt()
        e.exchange(count=False)
        lp.exchange(step)
    e.close()
    g, o = np.array(g_rows)[burn:], np.array(o_rows)[burn:]
    assert same >= 10, same                                 # identical counters for the first cycles
    n = g.shape[0]
    for col, name in ((1, 'pe'), (4, 'vol')):
        # slot statistics mix the replicas that visit the slot: compare slot means against the pooled scatter
        dm = g[:, :, col].mean(0) - o[:, :, col].mean(0)
        se = np.sqrt((g[:, :, col].var(0, ddof=1) + o[:, :, col].var(0, ddof=1)) / n)
        z = np.abs(dm) / np.maximum(se, 1e-12)
        assert np.median(z) < 1.5 and (z < 5.0).all(), (name, np.sort(z)[-4:])
        assert abs(dm.mean()) < 4.0 * np.sqrt((se ** 2).mean() / 64) + 1e-9, (name, dm.mean())
    # acceptance ratios (columns 14-16) averaged over the run and the adapted step sizes (5-7) at its end
    assert np.abs(g[:, :, 14:17].mean(0) - o[:, :, 14:17].mean(0)).max() < 0.2
    assert np.abs(g[:, :, 14:17].mean((0, 1)) - o[:, :, 14:17].mean((0, 1))).max() < 0.03
    ratio = g[-1, :, 5:8] / o[-1, :, 5:8]
    assert (ratio > 1 / 2.0).all() and (ratio < 2.0).all()
    assert abs(np.log(ratio).mean()) < 0.15


def test_launch_order_changes_nothing_but_the_schedule(monkeypatch):
    """More one-workgroup replicas than the chip has CUs (320 x 500 atoms, the kind of the reference's run.sh setting): from the second
    block on the workgroups are issued slowest-slot-first (nm_order_kernel, by the duration of each slot's previous block).  That is
    scheduling only: thermo rows, exchange permutations and configurations equal, bit for bit, those of the index-order launch."""
    import neuralmelting_amd as nm
    from neuralmelting_amd import lattice
    P = np.linspace(1.0, 8.0, 16, dtype=np.float32)
    T = np.linspace(0.25, 2.5, 20, dtype=np.float32)
    x, v, box, d = lattice.init_states(5, P, T, 0.03125, 0.03125)
    outs = []
    for order in ('0', '1'):
        monkeypatch.setenv('NM_LAUNCH_ORDER', order)
        e = nm.Engine(500, P, T)
        assert e.cus_per_replica == 1 and e.nslots == 320
        e.set_state(x, v, box, d)
        rows, perms = [], []
        for step in range(3):
            e.set_step(step)
            e.run_block(12)
            rows.append(e.thermo())
            e.adapt()
            e.exchange()
            perms.append(e.perm())
        outs.append((np.array(rows), np.array(perms), e.get_state()))
        e.close()
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    for a, b in zip(outs[0][2], outs[1][2]):
        np.testing.assert_array_equal(a, b)
