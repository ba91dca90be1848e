"""the .thrm / .traj reader (include/nm_parse.h, neuralmelting_amd/parse.py) against the reference's own numpy statements
(lammps_parse.py:48-49 and 88-96, restated below as the checker): bit-identical arrays, same files, shapes and dtypes"""
import os

import numpy as np
import pytest

from neuralmelting_amd import parse, remcmc

HEADER = '# ---\n# simulation parameters\n# ---\n# nsmpl: 4\n# | temp | pe | ke |\n'


def ref_thrm(path):
    return np.loadtxt(path, dtype=np.float32)                                    # parse:48-49


def ref_traj(path):
    with open(path, 'r') as f:                                                   # parse:88-94
        data = [line.split() for line in f.readlines()]
    natoms, box = np.split(np.array([v for v in data if len(v) == 2]), 2, 1)
    natoms = natoms.astype(np.uint16)[:, 0]
    box = box.astype(np.float32)[:, 0]
    x = [np.array(v).astype(np.float32) for v in data if len(v) == 3]
    return natoms, box, np.concatenate(tuple(x), 0).reshape(-1, 3)


def write_run(prefix, pn, tn, sn, natoms, rng, scale=1.0):
    """files laid out as consolidate_outputs leaves them (remcmc:297-310): replica-major, sn records each"""
    np.save(prefix + '.virial.trgt.npy', np.linspace(1, 8, pn, dtype=np.float32))
    np.save(prefix + '.temp.trgt.npy', np.linspace(0.25, 2.5, tn, dtype=np.float32))
    with open(prefix + '.thrm', 'w') as ft, open(prefix + '.traj', 'w') as fx:
        for k in range(pn * tn):
            ft.write(HEADER)
            for s in range(sn):
                row = rng.standard_normal(17) * 10.0 ** rng.integers(-3, 4, 17) * scale
                row[8:14] = rng.integers(0, 129, 6)
                ft.write(remcmc.Run.thrm_text(row))
                box = 6.0 + rng.random()
                fx.write(remcmc.Run.traj_text(natoms, box, (rng.random(3 * natoms) - 0.1) * box))


def test_thrm_matches_loadtxt(tmp_path):
    rng = np.random.default_rng(5)
    pre = str(tmp_path / 'a')
    write_run(pre, 2, 3, 4, 32, rng)
    for nthreads in (1, 3, 0):
        got = parse.read_thrm(pre + '.thrm', nthreads)
        want = ref_thrm(pre + '.thrm')
        assert got.dtype == np.float32 and got.shape == want.shape == (24, 17)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_traj_matches_reference_statements(tmp_path):
    rng = np.random.default_rng(6)
    pre = str(tmp_path / 'b')
    write_run(pre, 2, 2, 3, 108, rng)
    want = ref_traj(pre + '.traj')
    for nthreads in (1, 4, 0):
        got = parse.read_traj(pre + '.traj', nthreads)
        assert got[0].dtype == np.uint16 and got[1].dtype == np.float32 and got[2].dtype == np.float32
        for g, w in zip(got, want):
            assert g.shape == w.shape and g.tobytes() == w.tobytes()


def test_odd_tokens_take_the_slow_path(tmp_path):
    """values that are not '%.4E'-shaped (hand-edited files, NAN/INF rows, huge exponents, long mantissas)"""
    path = str(tmp_path / 'odd.thrm')
    toks = ['NAN', 'INF', '-INF', '1.0000E+300', '-1.0000E-300', '3.14159265358979', '1e-45', '7', '-0.0', '1.5E3', '+2.5000E+00',
            '9.9999E+22', '1.0000E-26', '0.0000E+00', '1.17549435E-38', '3.4028235E+38', '16777217']
    with open(path, 'w') as f:
        f.write('# header\n\n' + ' ' + ' '.join(toks) + '\n' + '\t'.join(toks[::-1]) + '  # trailing comment\n')
    got = parse.read_thrm(path, 2)
    with np.errstate(all='ignore'):
        want = np.array([[np.float32(float(t)) for t in toks], [np.float32(float(t)) for t in toks[::-1]]], np.float32)
    assert got.shape == (2, 17)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(got.view(np.uint32), np.loadtxt(path, dtype=np.float32).view(np.uint32))


def test_every_five_digit_mantissa_and_float32_ties(tmp_path):
    """the exact fast path over all mantissas and a spread of exponents, incl. values that sit on float32 rounding ties"""
    path = str(tmp_path / 'all.thrm')
    m = np.arange(10000, 100000)
    lines = []
    for e in (-26, -23, -22, -5, -1, 0, 3, 7, 18, 19, 22):
        vals = ['%d.%04dE%+03d' % (k // 10000, k % 10000, e) for k in m]
        vals += ['0.0000E+00'] * (-len(vals) % 17)
        lines += [' ' + ' '.join(vals[i:i + 17]) for i in range(0, len(vals), 17)]
    with open(path, 'w') as f:
        f.write('\n'.join(lines) + '\n')
    got = parse.read_thrm(path)
    want = np.array([float(t) for ln in lines for t in ln.split()], np.float64).astype(np.float32).reshape(-1, 17)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_errors_and_empty_files(tmp_path):
    with pytest.raises(RuntimeError, match='cannot open'):
        parse.read_thrm(str(tmp_path / 'missing.thrm'))
    bad = str(tmp_path / 'bad.thrm')
    open(bad, 'w').write(' 1.0 2.0 3.0\n')
    with pytest.raises(RuntimeError, match='17 columns'):
        parse.read_thrm(bad)
    empty = str(tmp_path / 'empty.traj')
    open(empty, 'w').close()
    n, b, x = parse.read_traj(empty)
    assert n.shape == (0,) and b.shape == (0,) and x.shape == (0, 3)
    nonl = str(tmp_path / 'nonl.traj')                       # last line without a newline; one stray line of 4 tokens is ignored
    open(nonl, 'w').write('2 6.1000E+00\n 1.0 2.0 3.0\n 1 2 3 4\n 4.0 5.0 6.5')
    n, b, x = parse.read_traj(nonl)
    assert n.tolist() == [2] and b.tolist() == [np.float32(6.1)] and x.tolist() == [[1, 2, 3], [4, 5, 6.5]]


def test_main_writes_the_reference_files(tmp_path, monkeypatch):
    rng = np.random.default_rng(7)
    monkeypatch.chdir(tmp_path)
    pre = str(tmp_path / 'run.lj.fcc.lammps')
    pn, tn, sn, na = 2, 4, 3, 32
    write_run(pre, pn, tn, sn, na, rng)
    parse.main(['-n', 'run', '-e', 'LJ'])
    cols = np.split(ref_thrm(pre + '.thrm'), 17, 1)                                # parse:48-66
    for name, col in zip(parse.COLUMNS, cols):
        a = np.load(pre + '.%s.npy' % name)
        assert a.dtype == np.float32 and a.shape == (pn, tn, sn)
        assert np.array_equal(a, col[:, 0].reshape(pn, tn, -1), equal_nan=True)
    natoms, box, x = ref_traj(pre + '.traj')
    a = np.load(pre + '.natoms.npy')
    assert a.dtype == np.uint16 and np.array_equal(a, natoms.reshape(pn, tn, -1))
    b = np.load(pre + '.box.npy')
    assert b.shape == (pn * tn * sn,) and np.array_equal(b, box)                   # flat: parse:92 never reshapes BOX
    c = np.load(pre + '.pos.npy')
    assert c.dtype == np.float32 and c.shape == (pn, tn, sn, na, 3) and np.array_equal(c, x.reshape(c.shape))


def test_large_file_is_cut_into_ranges(tmp_path):
    """several MiB so that the reader really runs multi-threaded ranges whose cuts fall mid-line"""
    rng = np.random.default_rng(8)
    path = str(tmp_path / 'big.traj')
    frames, na = 60, 2048
    with open(path, 'w') as f:
        for s in range(frames):
            f.write(remcmc.Run.traj_text(na, 12.0 + rng.random(), rng.standard_normal(3 * na) * 12.0))
    assert os.path.getsize(path) > 4 << 20
    n, b, x = parse.read_traj(path, 8)
    n1, b1, x1 = parse.read_traj(path, 1)
    assert n.tolist() == [na] * frames and x.shape == (frames * na, 3)
    assert np.array_equal(b, b1) and np.array_equal(x, x1)
    sample = rng.integers(0, frames * na, 2000)
    with open(path) as f:
        rows = [ln.split() for ln in f if len(ln.split()) == 3]
    for i in sample:
        assert np.array_equal(x[i], np.array(rows[i]).astype(np.float32))


def test_golden_from_the_reference_script(tmp_path, monkeypatch):
    """tests/golden/ref_parse.npz holds what the reference's own lammps_parse.py (run unmodified as a child process,
    tests/golden/make_golden_parse.py) wrote for the text stored next to it: all 20 files, bit for bit"""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_parse.npz'))
    monkeypatch.chdir(tmp_path)
    pre = str(tmp_path / 'g.lj.fcc.lammps')
    np.save(pre + '.virial.trgt.npy', g['P'])
    np.save(pre + '.temp.trgt.npy', g['T'])
    open(pre + '.thrm', 'wb').write(g['thrm_txt'].tobytes())
    open(pre + '.traj', 'wb').write(g['traj_txt'].tobytes())
    parse.main(['-n', 'g', '-e', 'LJ'])
    names = parse.COLUMNS + ('natoms', 'box', 'pos')
    assert len(names) == 20
    for n in names:
        got, want = np.load(pre + '.%s.npy' % n), g['ref_' + n]
        assert got.dtype == want.dtype and got.shape == want.shape, n
        assert got.tobytes() == want.tobytes(), n
    assert np.isnan(g['ref_virial']).sum() == 1          # the NAN row made it through both readers
