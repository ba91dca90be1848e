"""the driver (neuralmelting_amd.remcmc): flags, row partition, file formats, restart"""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import OracleEngine
from neuralmelting_amd import remcmc

REF_PARSE = '/root/reference/scripts/lammps_parse.py'


def test_parse_args_defaults_and_order():
    a = remcmc.parse_args([])
    # remcmc:89-100 order and remcmc:25-85 defaults
    assert a == (False, False, False, False, False, False, False, 128, 'remcmc_init', 1024, 'jobqueue', 'startup', 1, 20,
                 72, 32, 20, 1, 'fork', 'remcmc_init', 'LJ', 5, 16, 1, 8, 16, 0.25, 2.5, 0, 1024, 128, 0.125, 0.125, 8,
                 0.03125, 0.03125)
    a = remcmc.parse_args('-v -bm -n x -e LJ -ss 4 -pn 8 -pr 2 6 -tn 4 -tr 0.5 1.5 -sc 3 -sn 10 -sm 16 -pm 0.2 -vm 0.1 -ts 4 '
                          '-dx 0.01 -dv 0.02 -nw 16 -nt 1 -p -c -d -r -rd 5 -rn y -rs 7'.split())
    assert a[0] and a[6] and a[19] == 'x' and a[21] == 4 and a[22:28] == (8, 2.0, 6.0, 4, 0.5, 1.5)
    assert a[28:] == (3, 10, 16, 0.2, 0.1, 4, 0.01, 0.02) and a[7:10] == (5, 'y', 7)


@pytest.mark.parametrize('npn,world', [(8, 1), (8, 2), (8, 8), (16, 8), (5, 2), (2, 4)])
def test_row_partition(npn, world):
    rows = []
    for r in range(world):
        run = remcmc.Run(['-pn', str(npn), '-tn', '3', '-ss', '4'], rank=r, world=world)
        rows.extend(range(run.row0, run.row0 + run.nrows))
        assert run.k0 == run.row0 * 3 and run.nloc == run.nrows * 3
    assert rows == list(range(npn))          # contiguous, complete, whole pressure rows only


def parse_like_reference(pref, pn, tn):
    """what lammps_parse.py does with the two consolidated files (lammps_parse.py:44-63, 88-96)"""
    th = np.loadtxt(pref + '.thrm', dtype=np.float32)
    assert th.shape[1] == 17
    cols = [c[:, 0].reshape(pn, tn, -1) for c in np.split(th, 17, 1)]
    data = [ln.split() for ln in open(pref + '.traj')]
    two = np.array([v for v in data if len(v) == 2])
    natoms = two[:, 0].astype(np.uint16).reshape(pn, tn, -1)
    box = two[:, 1].astype(np.float32)
    x = np.concatenate([np.array(v).astype(np.float32) for v in data if len(v) == 3])
    x = x.reshape(pn, tn, natoms.shape[2], natoms[0, 0, 0], 3)
    return cols, natoms, box, x


def run_driver(tmp_path, argv, engine_factory=None):
    run = remcmc.Run(argv, cwd=str(tmp_path))
    if engine_factory is not None:
        run.make_engine = lambda: engine_factory(run)
    run.main()
    return run


def check_outputs(tmp_path, run, nrec):
    pref = run.PREF
    assert np.array_equal(np.load(pref + '.virial.trgt.npy'), run.P) and np.load(pref + '.virial.trgt.npy').dtype == np.float32
    assert np.array_equal(np.load(pref + '.temp.trgt.npy'), run.T)
    cols, natoms, box, x = parse_like_reference(pref, run.NP, run.NT)
    assert cols[0].shape == (run.NP, run.NT, nrec)
    # the library's reader (include/nm_parse.h) sees the same numbers as the reference's numpy statements
    from neuralmelting_amd import parse
    assert np.array_equal(parse.read_thrm(pref + '.thrm'), np.concatenate([c.reshape(-1, 1) for c in cols], 1), equal_nan=True)
    n2, b2, x2 = parse.read_traj(pref + '.traj')
    assert np.array_equal(n2, natoms.reshape(-1)) and np.array_equal(b2, box) and np.array_equal(x2, x.reshape(-1, 3))
    assert (natoms == run.natoms).all() and x.shape == (run.NP, run.NT, nrec, run.natoms, 3)
    assert np.isfinite(x).all() and (box > 5.0).all()
    # per-replica files are consolidated and removed (remcmc:314-316)
    import re
    assert not [f for f in os.listdir(tmp_path) if re.search(r'\.\d\d\.\d\d\.lammps\.(thrm|traj)$', f)]
    # restart dumps: STEP=-1 -> 0000 and every REFREQ (remcmc:975-976, 990-992)
    st = np.load(run.restart_file(run.NAME, 0), allow_pickle=True)
    assert st.shape == (run.NS, 21) and st[0][0] == run.natoms and len(st[0][1]) == 3 * run.natoms
    return cols


def test_driver_formats_with_oracle_engine(tmp_path, oracle):
    """host logic + file layout on a machine without a GPU: the engine is replaced by the oracle IN THIS TEST ONLY"""
    argv = '-bm -n t1 -e LJ -ss 4 -pn 2 -tn 2 -sn 3 -sm 4 -sc 1 -rd 2'.split()
    run = run_driver(tmp_path, argv, lambda r: OracleEngine(oracle, r))
    cols = check_outputs(tmp_path, run, nrec=2)          # STEP+1 > CUTOFF: cycles 2 and 3 are recorded
    assert os.path.isfile(run.restart_file('t1', 2))
    if os.path.isfile(REF_PARSE):
        # the reference's own consumer, unmodified, run as a script on the files just written
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
        subprocess.check_call([sys.executable, REF_PARSE, '-n', 't1', '-e', 'LJ'], cwd=str(tmp_path), env=env)
        pe = np.load(run.PREF + '.pe.npy')
        pos = np.load(run.PREF + '.pos.npy')
        assert pe.shape == (2, 2, 2) and pos.shape == (2, 2, 2, 256, 3)
        np.testing.assert_array_equal(pe, cols[1])


def test_cycles_that_write_nothing_go_to_the_engine_as_one_call(tmp_path, oracle):
    """-sc 5 of 8 cycles with a restart dump every 3: cycles 0-1 and 3-4 neither record nor dump and go out as run_cycles(2) each; the
    dump cycles (2, 5), the recorded ones (5, 6, 7) and the last go the single way — and every file equals the one of a driver whose engine
    has no run_cycles at all"""
    argv = '-bm -e LJ -ss 4 -pn 2 -tn 2 -sn 8 -sm 3 -sc 5 -rd 3'.split()
    calls = []

    class Counting(OracleEngine):
        def run_cycles(self, ncycles, mod):
            calls.append((self.step, ncycles))
            OracleEngine.run_cycles(self, ncycles, mod)

    class Plain(OracleEngine):
        run_cycles = property()     # hasattr() is False: the driver takes the cycle-by-cycle path

    a = tmp_path / 'a'; b = tmp_path / 'b'
    a.mkdir(); b.mkdir()
    ra = run_driver(a, argv + ['-n', 'q'], lambda r: Counting(oracle, r))
    rb = run_driver(b, argv + ['-n', 'q'], lambda r: Plain(oracle, r))
    assert calls == [(0, 2), (3, 2)]
    check_outputs(a, ra, nrec=3)
    for f in sorted(os.listdir(b)):
        fa, fb = os.path.join(a, f), os.path.join(b, f)
        if f.endswith('.npy'):
            xa, xb = np.load(fa, allow_pickle=True), np.load(fb, allow_pickle=True)
            if xa.dtype == object:
                assert all(np.array_equal(np.asarray(u), np.asarray(w)) for ra_, rb_ in zip(xa, xb) for u, w in zip(ra_, rb_)), f
            else:
                assert np.array_equal(xa, xb), f
        else:
            assert open(fa, 'rb').read() == open(fb, 'rb').read(), f


@pytest.mark.gpu
def test_driver_end_to_end_gpu(tmp_path):
    argv = '-bm -n g1 -e LJ -ss 4 -pn 2 -tn 4 -sn 4 -sm 8 -sc 0 -rd 2'.split()
    run = run_driver(tmp_path, argv)
    cols = check_outputs(tmp_path, run, nrec=4)
    temp, pe = cols[0], cols[1]
    assert (pe < 0).all() and (temp > 0).all()
    # restart from the dump of cycle 4 under a new name: loads, exchanges once, continues (remcmc:966-968)
    argv2 = '-r -rn g1 -rs 4 -bm -n g2 -e LJ -ss 4 -pn 2 -tn 4 -sn 2 -sm 8 -rd 2'.split()
    run2 = run_driver(tmp_path, argv2)
    check_outputs(tmp_path, run2, nrec=2)


@pytest.mark.gpu
def test_driver_with_a_sample_cutoff_writes_the_same_files_fused_or_not(tmp_path, monkeypatch):
    """-sc 8 of 12 cycles on the 8 x 8 grid, restart dump every 5: the cycles in front of the cutoff that neither record nor dump go to the engine as
    run_cycles (one launch each on this grid); every file the run leaves equals, byte for byte, the one of a run whose engine makes single launches of them"""
    argv = '-bm -e LJ -ss 4 -pn 8 -tn 8 -sn 12 -sm 8 -sc 8 -rd 5'.split()
    a = tmp_path / 'a'; b = tmp_path / 'b'
    a.mkdir(); b.mkdir()
    ra = run_driver(a, argv + ['-n', 'q'])
    monkeypatch.setenv('NM_FUSED_CYCLES', '0')
    run_driver(b, argv + ['-n', 'q'])
    monkeypatch.delenv('NM_FUSED_CYCLES')
    check_outputs(a, ra, nrec=4)
    names = sorted(os.listdir(b))
    assert names == sorted(os.listdir(a)) and any(n.endswith('.thrm') for n in names)
    for f in names:
        assert open(os.path.join(a, f), 'rb').read() == open(os.path.join(b, f), 'rb').read(), f


@pytest.mark.gpu
def test_pipeline_sampler_parse_distr_gpu(tmp_path, monkeypatch):
    """run.sh's first three stages (run.sh:7-16) on the build's own modules: sampler -> parse -> distr, file to file"""
    from neuralmelting_amd import distr, parse
    monkeypatch.chdir(tmp_path)
    run = run_driver(tmp_path, '-bm -n p1 -e LJ -ss 4 -pn 2 -tn 2 -sn 3 -sm 8'.split())
    parse.main(['-n', 'p1', '-e', 'LJ'])
    distr.main(['-n', 'p1', '-e', 'LJ', '-sb', '32', '-cb', '8'])
    pref = run.PREF
    pos = np.load(pref + '.pos.npy')
    assert pos.shape == (2, 2, 3, 256, 3) and pos.dtype == np.float32
    pe = np.load(pref + '.pe.npy')
    assert pe.shape == (2, 2, 3) and (pe < -1000).all()
    rdf = np.load(pref + '.rdf.npy')
    assert rdf.shape == (2, 2, 3, 32) and np.isfinite(rdf).all() and rdf[..., -4:].mean() > 0.5    # g(r) -> 1 at L/2
    assert rdf[..., :7].max() == 0.0                                                                # excluded core
    assert np.load(pref + '.cdf.npy').shape == (2, 2, 3, 8, 8, 8)


@pytest.mark.gpu
def test_driver_al_eam_gpu(tmp_path):
    """-e Al (BASELINE config 4 family): metal units, EAM kernel, same file layout"""
    argv = '-bm -n al1 -e Al -ss 4 -pn 2 -tn 2 -pr 1 8 -tr 300 900 -sn 2 -sm 8'.split()
    run = run_driver(tmp_path, argv)
    cols = check_outputs(tmp_path, run, nrec=2)
    assert (cols[0] > 50).all() and (cols[1] < -700).all()      # kelvin; ~ -3.3 eV/atom x 256


@pytest.mark.gpu
def test_driver_matches_oracle_driver(tmp_path, oracle):
    """same flags through the HIP engine and through the oracle stand-in give the same .thrm rows (5 significant digits)"""
    argv = '-bm -e LJ -ss 4 -pn 2 -tn 2 -sn 3 -sm 6'.split()
    a = tmp_path / 'a'; b = tmp_path / 'b'
    a.mkdir(); b.mkdir()
    ra = run_driver(a, argv + ['-n', 'a'])
    rb = run_driver(b, argv + ['-n', 'a'], lambda r: OracleEngine(oracle, r))
    ta = np.loadtxt(ra.PREF + '.thrm'); tb = np.loadtxt(rb.PREF + '.thrm')
    np.testing.assert_allclose(ta, tb, rtol=2e-4, atol=1e-4)     # the text carries 5 digits
    np.testing.assert_array_equal(ta[:, 8:14], tb[:, 8:14])


def test_restart_file_written_by_the_reference_loads(tmp_path):
    """tests/golden/ref_restart.npz holds the bytes of a restart file written by the REFERENCE's own dump_samples_restart
    (remcmc:821-828, tests/golden/make_golden_restart.py): the driver's -r path reads exactly the numbers that went in, and
    what the driver dumps has the same layout (object array NS x 21, same entry types)"""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_restart.npz'))
    run = remcmc.Run('-r -rn gold -rs 7 -n new -e LJ -ss 2 -pn 2 -tn 2'.split(), cwd=str(tmp_path))
    open(run.restart_file('gold', 7), 'wb').write(g['rstrt'].tobytes())
    x, v, box, d, th = run.load_samples_restart()
    for got, want in ((x, g['x']), (v, g['v']), (box, g['box']), (d, g['d']), (th, g['th'])):
        assert got.dtype == np.float64 and np.array_equal(got, want)
    ref = list(np.load(run.restart_file('gold', 7), allow_pickle=True))
    # the driver's own dump of the same state: same shape, same kinds of entries
    class Eng:
        nslots, natoms = 4, 32
        def get_state(self, **k): return x, v, box, d
        def thermo(self):
            rows = np.zeros((4, 17)); rows[:, :5] = th[:, [0, 1, 2, 3, 4]]; rows[:, 5:8] = d
            return rows
    run.engine = Eng()
    run.STEP = 0
    run.dump_samples_restart()
    mine = list(np.load(run.restart_file('new', 1), allow_pickle=True))
    assert len(mine) == len(ref) == 4
    for a, b in zip(mine, ref):
        assert len(a) == len(b) == 21 and a[0] == b[0] == 32
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        assert [float(a[i]) for i in range(3, 12)] == [float(b[i]) for i in range(3, 12)]
        assert [float(q) for q in a[12:]] == [0.0] * 9 == [float(q) for q in b[12:]]      # counters and ratios are zero after gen_mc_param


def test_output_files_equal_the_references(tmp_path):
    """tests/golden/ref_outputs.npz: the consolidated .thrm / .traj the REFERENCE's own init_outputs, init_headers,
    write_outputs and consolidate_outputs leave for three recorded cycles of a 2x3 grid (tests/golden/make_golden_outputs.py).
    The driver's writer (headers in Python, rows and frames through nm_append_outputs, chunked consolidation) must leave the
    same bytes."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_outputs.npz'))
    npn, ntn, sz, ncyc = [int(q) for q in g['shape']]
    run = remcmc.Run(('-n gout -e LJ -ss %d -pn %d -tn %d -sn 5 -sc 2 -sm 16' % (sz, npn, ntn)).split(), cwd=str(tmp_path))
    run.init_outputs()
    run.init_headers()
    for c in range(ncyc):
        run.write_outputs(g['rows'][c], g['x'][c], g['box'][c])
    run.consolidate_outputs()
    assert open(run.PREF + '.thrm', 'rb').read() == g['thrm'].tobytes()
    assert open(run.PREF + '.traj', 'rb').read() == g['traj'].tobytes()
    assert sorted(os.listdir(tmp_path)) == ['gout.lj.fcc.lammps.thrm', 'gout.lj.fcc.lammps.traj']
