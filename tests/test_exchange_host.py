"""the host-side sweep used when a pressure row spans ranks (neuralmelting_amd/exchange.py) draws and decides exactly like the
oracle's (and therefore the device's) sweep"""
import numpy as np
import pytest

from neuralmelting_amd import exchange as X


def test_python_philox_known_answers():
    assert X.philox4x32_10((0, 0, 0, 0), (0, 0)) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    assert X.philox4x32_10((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == \
        (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)


@pytest.mark.parametrize('npn,nt', [(1, 4), (2, 8), (3, 5)])
def test_host_sweep_equals_oracle(oracle, npn, nt):
    rng = np.random.default_rng(npn * 10 + nt)
    ns = npn * nt
    T = np.linspace(0.25, 2.5, nt)
    et = np.tile(T, npn); pf = np.repeat(np.linspace(1, 8, npn), nt) / et
    for step in range(5):
        etot = -1900.0 + 3.0 * rng.random(ns); vol = 225.0 + rng.random(ns)
        swaps, perm, _, _, _ = oracle.exchange(npn, nt, 0, npn, 256, step, etot, vol, et, pf)
        p2, s2 = X.sweep(npn, nt, 256, step, etot, vol, et, pf)
        assert s2 == swaps and list(p2) == list(perm)


@pytest.mark.gpu
def test_host_sweep_equals_device_kernel():
    import neuralmelting_amd as nm
    P = np.linspace(1, 8, 2, dtype=np.float32); T = np.linspace(0.25, 2.5, 8, dtype=np.float32)
    e = nm.Engine(256, P, T)
    rng = np.random.default_rng(9)
    th = np.zeros((16, 5)); th[:, 1] = -1900 + 3 * rng.random(16); th[:, 2] = 100 + rng.random(16); th[:, 4] = 225 + rng.random(16)
    e.set_thermo(th); e.set_step(11)
    n = e.exchange()
    et, pf = e.constants()
    perm, swaps = X.sweep(2, 8, 256, 11, th[:, 1] + th[:, 2], th[:, 4], et, pf)
    assert swaps == n and list(perm) == list(e.perm())
    # a context holding half a row refuses the device sweep
    e2 = nm.Engine(256, P, T, slot0=4, nslots=4)
    assert e2.nslots == 4
    with pytest.raises(nm.NMError):
        e2.exchange()
    e2.close(); e.close()
