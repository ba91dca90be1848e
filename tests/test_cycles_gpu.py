"""nm_run_cycles: `ncycles` cycles of the main loop with outputs off (remcmc:977-995) as ONE launch in which only the replicas of a pressure
row wait for one another.  The bar: the same chains, bit for bit, as the loop of nm_run_block + nm_adapt + nm_exchange — on every
configuration that has the fused kernel, on those that fall back to the loop, against the oracle's loop, and through a healed launch."""
import numpy as np
import pytest

from helpers import OracleLoop, grids

pytestmark = pytest.mark.gpu

RTOL = 1e-6


def _engine(sz, P, T, el, x, v, box, d):
    import neuralmelting_amd as nm
    e = nm.Engine(4 * sz ** 3, P, T, element=el)
    e.set_state(x, v, box, d)
    return e


def _single(e, step0, ncyc, mod):
    for s in range(step0, step0 + ncyc):
        e.set_step(s)
        e.run_block(mod)
        e.adapt()
        e.exchange(count=False)


def _everything(e):
    e.synchronize()
    x, v, box, d = e.get_state()
    return dict(x=x, v=v, box=box, d=d, thermo=e.thermo(), perm=e.perm(), status=e.status())


# (element, sz, pressure rows, temperatures, workgroups per replica the grid gets, has a fused kernel?)
# The fused kernel is instantiated for 4^3 at 2, 4, 8 workgroups per replica and 6^3 at 8; nm_run_cycles uses it where it measured faster (4^3 at 2
# and 4), everywhere under NM_FUSED_CYCLES=all — what the comparison below sets, so that every instantiation is checked — and nowhere under =0.
CASES = [('LJ', 4, 4, 8, 8, True), ('LJ', 4, 8, 8, 4, True), ('LJ', 4, 16, 8, 2, True), ('LJ', 4, 32, 8, 1, False),
         ('LJ', 4, 2, 16, 8, True), ('LJ', 4, 3, 5, 8, True),
         ('Al', 4, 8, 8, 4, True), ('Al', 4, 16, 8, 2, True),
         ('LJ', 6, 4, 8, 8, True), ('LJ', 6, 8, 8, 4, False)]


@pytest.mark.parametrize('el,sz,npn,ntn,cus,fused', CASES)
def test_fused_cycles_equal_the_loop_of_single_calls(monkeypatch, el, sz, npn, ntn, cus, fused):
    from neuralmelting_amd import lattice
    monkeypatch.setenv('NM_FUSED_CYCLES', 'all')
    pr, tr = ((1.0, 8.0), (0.25, 2.5)) if el == 'LJ' else ((1.0, 8.0), (256.0, 2560.0))
    P, T = grids(npn, ntn, pr, tr)
    x, v, box, d = lattice.init_states(sz, P, T, 0.03125, 0.03125, el=el)
    ncyc, mod = (5, 12) if sz == 4 else (3, 6)
    a = _engine(sz, P, T, el, x, v, box, d)
    assert a.cus_per_replica == cus
    a.timing_reset()
    a.set_step(7)
    a.run_cycles(ncyc, mod)
    got = _everything(a)
    launches = a.timing()[0]
    assert launches == (1 if fused else ncyc)       # one launch where the configuration has the kernel, the loop elsewhere
    # and a second call goes on from where the first left the chains (steps 7 + ncyc ..)
    a.set_step(7 + ncyc)
    a.run_cycles(2, mod)
    got2 = _everything(a)
    assert a.note() == '' and a.heals == 0
    a.close()
    b = _engine(sz, P, T, el, x, v, box, d)
    _single(b, 7, ncyc, mod)
    want = _everything(b)
    _single(b, 7 + ncyc, 2, mod)
    want2 = _everything(b)
    b.close()
    for g, w in ((got, want), (got2, want2)):
        assert (g['status'] == 0).all()
        for key in w:
            np.testing.assert_array_equal(g[key], w[key], err_msg=key)
    if el == 'LJ':
        assert not np.array_equal(want2['perm'], np.arange(npn * ntn))   # the rows did exchange: the comparison is not vacuous


def test_which_grids_run_their_cycles_as_one_launch():
    """one launch for the 4^3 clusters of 2 and 4 workgroups by default, for every configuration that has the kernel under NM_FUSED_CYCLES=all,
    the loop of single launches under =0 and wherever there is no such kernel"""
    import neuralmelting_amd as nm
    import os
    for env, ones in ((None, (4, 2)), ('0', ()), ('all', (4, 2, 8))):
        for npn, ntn, cus in ((8, 8, 4), (16, 8, 2), (4, 8, 8), (32, 8, 1)):
            if env is None:
                os.environ.pop('NM_FUSED_CYCLES', None)
            else:
                os.environ['NM_FUSED_CYCLES'] = env
            try:
                P, T = grids(npn, ntn)
                e = nm.Engine(256, P, T)
                e.init_lattice()
                assert e.cus_per_replica == cus
                e.timing_reset()
                e.run_cycles(3, 4)
                e.synchronize()
                assert e.timing()[0] == (1 if cus in ones else 3), (env, cus)
                e.close()
            finally:
                os.environ.pop('NM_FUSED_CYCLES', None)


def test_a_long_call_is_cut_into_launches_of_at_most_64_cycles():
    import neuralmelting_amd as nm
    from neuralmelting_amd import lattice
    P, T = grids(8, 8)
    x, v, box, d = lattice.init_states(4, P, T, 0.03125, 0.03125)
    a = nm.Engine(256, P, T)
    a.set_state(x, v, box, d)
    a.timing_reset()
    a.set_step(0)
    a.run_cycles(150, 1)
    got = _everything(a)
    assert a.timing()[0] == 3                                        # 64 + 64 + 22
    a.close()
    b = nm.Engine(256, P, T)
    b.set_state(x, v, box, d)
    _single(b, 0, 150, 1)
    want = _everything(b)
    b.close()
    for key in want:
        np.testing.assert_array_equal(got[key], want[key], err_msg=key)


def test_fused_cycles_against_the_oracles_main_loop(oracle, monkeypatch):
    """the oracle's gen_samples / gen_mc_params / replica_exchange, four cycles, against one fused launch"""
    monkeypatch.setenv('NM_FUSED_CYCLES', 'all')
    sz, mod, ncyc = 4, 8, 4
    P, T = grids(3, 4)
    loop = OracleLoop(oracle, sz, P, T)
    import neuralmelting_amd as nm
    e = nm.Engine(4 * sz ** 3, P, T, seed=loop.seed)
    e.set_state(loop.x, loop.v, loop.box, loop.d)
    assert e.cus_per_replica == 8
    e.timing_reset()
    e.set_step(0)
    e.run_cycles(ncyc, mod)
    e.synchronize()
    assert e.timing()[0] == 1
    for step in range(ncyc):
        loop.run_block(mod, step)
        loop.adapt()
        loop.exchange(step)
    x, v, box, d = e.get_state()
    np.testing.assert_array_equal(d, loop.d)                         # step sizes: the same accept / reject history and the same swaps
    np.testing.assert_allclose(box, loop.box, rtol=1e-12)
    np.testing.assert_allclose(x, loop.x, rtol=0, atol=1e-7)
    th = e.thermo()
    np.testing.assert_allclose(th[:, :5], loop.thermo, rtol=RTOL)
    assert (th[:, 8:] == 0).all()                                    # gen_mc_params ran after the last block
    e.close()


def test_run_cycles_needs_whole_pressure_rows_and_a_positive_count():
    import neuralmelting_amd as nm
    from neuralmelting_amd import _lib as B
    P, T = grids(2, 4)
    e = nm.Engine(256, P, T, slot0=2, nslots=4)                      # half of row 0 and half of row 1
    with pytest.raises(nm.NMError) as ei:
        e.run_cycles(2, 4)
    assert ei.value.code == B.NM_ERR_UNSUPPORTED
    e.close()
    e = nm.Engine(256, P, T)
    for bad in (0, -1):
        with pytest.raises(nm.NMError) as ei:
            e.run_cycles(bad, 4)
        assert ei.value.code == B.NM_ERR_ARG
    e.close()


def test_a_fused_launch_that_is_not_resident_is_reissued_whole(monkeypatch):
    """The residency census is taken once, in the first block of the fused launch, before anything is touched: failing it
    (NM_INJECT_CENSUS) stops every cycle of the launch, and the next host call re-issues all of them at fewer workgroups per
    replica — the chains are those of a context that ran at 2 per replica from the start."""
    import neuralmelting_amd as nm
    from neuralmelting_amd import lattice
    P, T = grids(8, 8)
    x, v, box, d = lattice.init_states(4, P, T, 0.03125, 0.03125)
    ncyc, mod = 4, 12
    monkeypatch.setenv('NM_INJECT_CENSUS', '0')
    a = nm.Engine(256, P, T)
    assert a.cus_per_replica == 4
    a.set_state(x, v, box, d)
    a.set_step(3)
    a.run_cycles(ncyc, mod)
    a.set_step(3 + ncyc)
    a.run_block(mod)                                                 # queued behind the launch that fails: must not run before the re-issue
    got = _everything(a)
    monkeypatch.delenv('NM_INJECT_CENSUS')
    assert a.cus_per_replica == 2 and a.heals == 1
    assert 'stopped at 4 workgroups per replica (grid not resident); re-issued at 2' in a.note()
    a.close()
    monkeypatch.setenv('NM_CUS_PER_REPLICA', '2')
    b = nm.Engine(256, P, T)
    b.set_state(x, v, box, d)
    _single(b, 3, ncyc, mod)
    b.set_step(3 + ncyc)
    b.run_block(mod)
    want = _everything(b)
    b.close()
    for key in want:
        np.testing.assert_array_equal(got[key], want[key], err_msg=key)


def test_an_error_inside_a_fused_launch_is_loud_and_the_grid_drains(monkeypatch):
    """A replica that stops on an error in the middle of a fused launch (an injected list overflow) sets the launch's abort word: the other workgroups
    of its row do not sit out the 2 s of a hand-over timeout, the other rows leave at their next barrier, the host gets NM_ERR_STATE once with the slot
    and the reason — and the context is usable again from a state the caller sets."""
    import time
    import neuralmelting_amd as nm
    from neuralmelting_amd import _lib as B
    from neuralmelting_amd import lattice
    P, T = grids(8, 8)
    x, v, box, d = lattice.init_states(4, P, T, 0.03125, 0.03125)
    e = nm.Engine(256, P, T)
    e.set_state(x, v, box, d)
    e.run_block(8)
    e.synchronize()
    monkeypatch.setenv('NM_INJECT_OVERFLOW', '3,1')
    e.timing_reset()
    e.set_step(1)
    t0 = time.perf_counter()
    e.run_cycles(6, 48)
    e.set_step(7)
    e.run_block(48)
    with pytest.raises(nm.NMError) as ei:
        e.synchronize()
    assert time.perf_counter() - t0 < 1.0                             # six cycles are ~35 ms; a timeout would be 2 s
    monkeypatch.delenv('NM_INJECT_OVERFLOW')
    assert ei.value.code == B.NM_ERR_STATE and 'neighbour list overflow' in str(ei.value)
    st = e.status()
    assert (st != 0).any() and (st[st != 0] == 1).all()               # NM_ST_LIST_OVERFLOW and nothing else: nobody timed out waiting
    # usable again: the same three cycles as a fresh context
    e.adapt()                                                         # (zero the counters and ratios the stopped launch left ...
    e.set_state(x, v, box, d)                                         #  ... then the state, step sizes included)
    e.set_step(0)
    e.run_cycles(3, 8)
    got = _everything(e)
    cus_after, heals_after = e.lib.nm_cus_per_replica(e.h), e.heals
    e.close()
    f = nm.Engine(256, P, T)
    f.set_state(x, v, box, d)
    f.set_step(0)
    f.run_cycles(3, 8)
    want = _everything(f)
    f.close()
    assert (got['status'] == 0).all() and cus_after == 4 and heals_after == 0   # (the launches behind the error left the census counters in step)
    for key in ('x', 'v', 'box', 'd'):                                 # (not the slot-to-buffer map: rows that got ahead of the error had exchanged)
        np.testing.assert_array_equal(got[key], want[key], err_msg=key)
