"""GPU tests against the goldens produced by the REFERENCE's own functions (tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
G = json.load(open(os.path.join(HERE, 'golden', 'ref_scalars.json')))
B = np.load(os.path.join(HERE, 'golden', 'ref_blocks.npz'))
RTOL = 1e-6


@pytest.mark.parametrize('tag', ['lj_2x2', 'lj_8x8'])
def test_G1_constants_engine(tag):
    import neuralmelting_amd as nm
    g = G['G1_constants'][tag]
    e = nm.Engine(256, np.float32(g['P']), np.float32(g['T']))
    et, pf = e.constants()
    np.testing.assert_array_equal(et, g['et'])
    np.testing.assert_array_equal(pf, g['pf'])
    e.close()


def test_G1_constants_engine_al():
    """init_constant in metal units (remcmc:124-127) as the reference computed it for the 8x8 Al grid"""
    import neuralmelting_amd as nm
    g = G['G1_constants']['al_8x8']
    e = nm.Engine(256, np.float32(g['P']), np.float32(g['T']), element='Al')
    et, pf = e.constants()
    np.testing.assert_array_equal(et, g['et'])
    np.testing.assert_array_equal(pf, g['pf'])
    e.close()


def test_G2_adapt_engine():
    """nm_adapt_kernel on the reference's own gen_mc_param cases (remcmc:726-745): every golden case sits in one slot of a grid,
    is handed its float32 ratios and step sizes, and must come back with the reference's step sizes, zeroed counters and ratios"""
    import neuralmelting_amd as nm
    cases = G['G2_adapt']
    n = len(cases)
    nt = n
    P = np.linspace(1, 8, 1, dtype=np.float32)
    T = np.linspace(0.25, 2.5, nt, dtype=np.float32)
    e = nm.Engine(256, P, T)
    steps = np.array([c['steps_in'] for c in cases])
    e.set_state(dxdvdt=steps)
    ratios = np.array([[0.0 if r != r else r for r in c['ratios']] for c in cases], dtype=np.float32)   # nan_to_num, remcmc:686-688
    e.set_counters(count=np.full((n, 6), 7.0), ratio=ratios)
    e.adapt()
    rows = e.thermo()
    np.testing.assert_array_equal(rows[:, 5:8], np.array([c['steps_out'] for c in cases]))
    np.testing.assert_array_equal(rows[:, 8:], 0.0)                                                # counters and ratios zeroed
    e.close()


def _count_swaps(g):
    """replica_exchange (remcmc:776-803) on a G3 golden's inputs and recorded uniforms: (accepted swaps, final arrangement)"""
    npn, nt = g['np'], g['nt']
    etot = np.array(g['pe']) + np.array(g['ke'])
    vol = np.array(g['vol'])
    ident = list(range(npn * nt))      # which configuration sits in each slot
    u = iter(g['uniforms'])
    n = 0
    for r in range(npn):
        for v in range(nt - 1, -1, -1):
            for w in range(v):
                i, j = r * nt + v, r * nt + w
                a, b = ident[i], ident[j]
                dh = (etot[a] - etot[b]) * (1.0 / g['et'][i] - 1.0 / g['et'][j]) + (g['pf'][i] - g['pf'][j]) * (vol[a] - vol[b])
                if next(u) <= min(1.0, np.exp(dh)):
                    ident[i], ident[j] = b, a
                    n += 1
    return n, ident


@pytest.mark.parametrize('idx', range(4))
def test_G3_exchange_engine(idx):
    """the device sweep, fed the uniforms the reference drew, ends in the reference's arrangement"""
    import neuralmelting_amd as nm
    g = G['G3_exchange'][idx]
    P = np.linspace(1, 8, g['np'], dtype=np.float32)
    T = np.linspace(0.25, 2.5, g['nt'], dtype=np.float32)
    e = nm.Engine(256, P, T)
    ns = g['np'] * g['nt']
    th = np.zeros((ns, 5))
    th[:, 1], th[:, 2], th[:, 4] = g['pe'], g['ke'], g['vol']
    e.set_thermo(th)
    e.set_exchange_tape(g['uniforms'])
    nsw = e.exchange()
    assert list(e.perm()) == g['perm']
    # the number of accepted swaps: the sweep of remcmc:782-798 replayed here on the golden's own energies, volumes, constants and
    # uniforms (it must end in the golden's arrangement, which pins the replay), counted pair by pair
    want, perm = _count_swaps(g)
    assert perm == g['perm']
    assert nsw == want
    # a second sweep without the tape must leave the tape path (Philox) — just exercise it
    e.set_exchange_tape(None)
    e.exchange()
    e.close()


@pytest.mark.parametrize('tag', ['bulk', 'iter', 'default_mix'])
def test_G5_blocks_engine(tag):
    """nm_run_block replays the reference's recorded np.random stream and lands on the reference's gen_sample output"""
    import neuralmelting_amd as nm
    mod, ppos, pvol, nstps, bm = B[tag + '_params']
    P = np.linspace(1, 8, 2, dtype=np.float32)
    T = np.linspace(0.25, 2.5, 2, dtype=np.float32)
    e = nm.Engine(256, P, T, ppos=ppos, pvol=pvol, nstps=int(nstps), bulk=bool(bm), seed=256)
    x = np.array([B['%s_%d_x_in' % (tag, k)] for k in range(4)])
    v = np.array([B['%s_%d_v_in' % (tag, k)] for k in range(4)])
    sc = np.array([B['%s_%d_scal_in' % (tag, k)] for k in range(4)])
    e.set_state(x, v, sc[:, 0], sc[:, 1:4])
    et, pf = e.constants()
    np.testing.assert_array_equal(et, sc[:, 4])
    np.testing.assert_array_equal(pf, sc[:, 5])
    e.set_rng_tape([B['%s_%d_tape' % (tag, k)] for k in range(4)])
    e.set_step(3)
    e.run_block(int(mod))
    rows = e.thermo()
    xo, vo, boxo, _ = e.get_state()
    for k in range(4):
        pre = '%s_%d_' % (tag, k)
        row = B[pre + 'row_out']
        np.testing.assert_array_equal(rows[k, 8:14], row[8:14])                     # counters
        np.testing.assert_array_equal(rows[k, 14:17].astype(np.float32), row[14:17].astype(np.float32))
        np.testing.assert_allclose(rows[k, :5], row[:5], rtol=RTOL)
        assert boxo[k] == B[pre + 'box_out'][0]
        np.testing.assert_allclose(xo[k], B[pre + 'x_out'], rtol=0, atol=1e-8)
        np.testing.assert_allclose(vo[k], B[pre + 'v_out'], rtol=0, atol=1e-7)
    e.close()
