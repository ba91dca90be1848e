"""the library's host-side '%.4E' formatter is byte-identical to Python's formatting (what the reference writes)"""
import ctypes as C

import numpy as np

from neuralmelting_amd import _lib, remcmc


def py_thrm(row):
    return 17 * ' %.4E' % tuple(row) + '\n'               # remcmc:245


def py_traj(natoms, box, x):
    out = '%d %.4E\n' % (natoms, box)                       # remcmc:254
    for i in range(natoms):
        out += 3 * ' %.4E' % tuple(x[3 * i:3 * i + 3]) + '\n'   # remcmc:256
    return out


def test_e4_random_and_special_values():
    rng = np.random.default_rng(1)
    vals = np.concatenate([
        rng.standard_normal(200000) * 10.0 ** rng.integers(-30, 30, 200000),
        rng.random(100000) * 7.0, -rng.random(50000) * 2500.0,
        np.float64(np.arange(0, 100000)) * 1e-4,                       # exact decimal-looking values
        np.array([0.0, -0.0, 1.0, 9.99995, 9.999949999, 0.99995, 99999.5, 1e22, 1e23, 1e-22, 1e-23, 5e-324, 1.7976931348623157e308,
                  1.00005, 1.00015, 1.00025, 2.5, 0.125, 12345.5, 1234.55, 123.455, np.inf, -np.inf, 6.0303160521, 128.0, 71.0 / 99.0])])
    vals = vals[:len(vals) - len(vals) % 17]
    for blk in vals.reshape(-1, 17)[::1]:
        assert remcmc.Run.thrm_text(blk) == py_thrm(blk)
    # midpoints of the fifth digit that are exactly representable: ties must go to even like printf does
    ties = np.array([1.00005, 1.00015, 1.00025, 1.00035]) * 1.0
    exact = np.array([0.5 + k for k in range(10000, 10017)], dtype=np.float64)   # 10000.5 ... exact binary ties at 5 digits
    assert remcmc.Run.thrm_text(exact) == py_thrm(exact)
    assert remcmc.Run.thrm_text(np.resize(ties, 17)) == py_thrm(np.resize(ties, 17))


def test_traj_frame_and_files(tmp_path):
    rng = np.random.default_rng(2)
    n = 256
    x = rng.random((3, 3 * n)) * 6.2 - 0.01
    box = np.array([6.1703858, 6.0303161, 6.4281749])
    for k in range(3):
        assert remcmc.Run.traj_text(n, box[k], x[k]) == py_traj(n, box[k], x[k])
    rows = rng.standard_normal((3, 17)) * 100
    L = _lib.load()
    thrm = [str(tmp_path / ('r%d.thrm' % k)).encode() for k in range(3)]
    traj = [str(tmp_path / ('r%d.traj' % k)).encode() for k in range(3)]
    for rep in range(2):   # append mode
        rc = L.nm_append_outputs(3, n, (C.c_char_p * 3)(*thrm), (C.c_char_p * 3)(*traj), rows.ctypes.data_as(_lib.c_double_p),
                                 x.ctypes.data_as(_lib.c_double_p), box.ctypes.data_as(_lib.c_double_p), 2)
        assert rc == 0
    for k in range(3):
        assert open(thrm[k].decode()).read() == 2 * py_thrm(rows[k])
        assert open(traj[k].decode()).read() == 2 * py_traj(n, box[k], x[k])
