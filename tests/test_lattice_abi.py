"""nm_lattice_state / nm_init_lattice (init_sample, remcmc:394-433, behind the C-ABI for callers without the Python front end)
against neuralmelting_amd/lattice.py, the front end's own initial-state generator (SciPy root finder, numpy Philox)."""
import ctypes as C

import numpy as np
import pytest

from neuralmelting_amd import _lib as B
from neuralmelting_amd import lattice


def abi_state(el, sz, P, nt, seed, gslot, dx, interpolate):
    L = B.load()
    n = 4 * sz ** 3
    x = np.empty(3 * n)
    box = C.c_double(0.0)
    P = np.ascontiguousarray(P, dtype=np.float32)
    rc = L.nm_lattice_state({'LJ': 0, 'Al': 1}[el], sz, len(P), nt, P.ctypes.data_as(B.c_float_p), seed, gslot, dx, int(interpolate),
                            x.ctypes.data_as(B.c_double_p), C.byref(box))
    assert rc == 0
    return x, box.value


@pytest.mark.parametrize('el,sz,interp', [('LJ', 4, False), ('LJ', 4, True), ('LJ', 6, False), ('Al', 4, False)])
def test_lattice_state_equals_the_python_front_end(el, sz, interp):
    P = np.linspace(1.0, 8.0, 3, dtype=np.float32)
    T = np.linspace(0.25, 2.5, 4, dtype=np.float32) if el == 'LJ' else np.linspace(300.0, 900.0, 4, dtype=np.float32)
    x, v, box, d = lattice.init_states(sz, P, T, 0.03125, 0.03125, el=el, seed=256, interpolate=interp)
    for g in (0, 5, 11):
        xa, ba = abi_state(el, sz, P, 4, 256, g, 0.03125, interp)
        assert abs(ba - box[g]) <= 1e-11 * box[g]                       # two root finders on the same smooth function
        # identical draws (numpy's Philox4x64 keyed [seed, slot]); coordinates differ only through the box edge.  An atom that sits
        # within that difference of a face may wrap on one side only: compare modulo the box
        dd = xa - x[g]
        dd -= ba * np.rint(dd / ba)
        assert np.abs(dd).max() < 1e-9
    if el == 'LJ' and sz == 4 and not interp:                            # relaxed 4^3 edges at P* = 1 and 8
        assert abs(abi_state('LJ', 4, [1.0], 1, 256, 0, 0.0, False)[1] - 6.170385810) < 2e-9
        assert abs(abi_state('LJ', 4, [8.0], 1, 256, 0, 0.0, False)[1] - 6.030316052) < 2e-9


def test_bad_arguments_are_refused():
    L = B.load()
    x = np.empty(768)
    box = C.c_double(0.0)
    P = np.float32([1.0])
    assert L.nm_lattice_state(7, 4, 1, 1, P.ctypes.data_as(B.c_float_p), 1, 0, 0.03, 0, x.ctypes.data_as(B.c_double_p), C.byref(box)) == B.NM_ERR_ARG
    assert L.nm_lattice_state(0, 4, 1, 1, P.ctypes.data_as(B.c_float_p), 1, 5, 0.03, 0, x.ctypes.data_as(B.c_double_p), C.byref(box)) == B.NM_ERR_ARG


@pytest.mark.gpu
def test_init_lattice_fills_the_context_like_set_state(oracle=None):
    import neuralmelting_amd as nm
    P = np.linspace(1.0, 8.0, 2, dtype=np.float32)
    T = np.linspace(0.25, 2.5, 4, dtype=np.float32)
    x, v, box, d = lattice.init_states(4, P, T, 0.03125, 0.03125)
    outs = []
    for use_abi in (True, False):
        e = nm.Engine(256, P, T)
        if use_abi:
            e.init_lattice(0.03125, 0.03125)
        else:
            e.set_state(x, v, box, d)
        xs, vs, bs, ds = e.get_state()
        e.run_block(8)
        outs.append((xs, vs, bs, ds, e.thermo()))
        e.close()
    a, b = outs
    np.testing.assert_allclose(a[2], b[2], rtol=1e-11)
    np.testing.assert_array_equal(a[3], b[3])
    np.testing.assert_array_equal(a[1], 0.0)
    np.testing.assert_array_equal(a[4][:, 8:14], b[4][:, 8:14])          # eight moves later: same decisions
    np.testing.assert_allclose(a[4][:, :5], b[4][:, :5], rtol=1e-6)
