"""Generates the golden fixtures of tests/golden/ by running the REFERENCE's own Python functions
(/root/reference/scripts/lammps_remcmc.py, "remcmc") in this container.  Run once here; only the numbers it
writes travel (the reference itself never leaves this container and is never copied).

The reference cannot be imported as is: `numba` and `lammps` are not installed (ordinary ModuleNotFoundError) and
`np.unravel_index(k, dims=...)` lost its `dims` keyword in NumPy 2.  So, as SURVEY.md §8c describes:
  * `numba` is replaced by an in-process stub whose `jit` is the identity (the reference's only use, remcmc:776,
    falls back to plain Python under numba as well);
  * `lammps.lammps` is replaced by FakeLammps below: it accepts exactly the method calls / command strings the
    reference issues and answers them with the CPU oracle's primitives.  The goldens
    therefore pin everything the REFERENCE contributes — move selection, Metropolis criteria, '%f' command
    strings, counter logic, RNG-consumption order, exchange sweep, adaptive steps, file formats — but NOT the
    arithmetic of LAMMPS itself (no liblammps exists here: "parity unpinned" for that part);
  * P and T are handed to the module as float64 arrays holding the float32-rounded grid values, which
    reproduces the NumPy-1.x promotion the reference was written for.

Outputs: ref_scalars.json (G1 constants, G2 adaptive steps, G3 exchange sweeps, G4 strings) and
ref_blocks.npz (G5: gen_sample traces: inputs, the recorded np.random stream, outputs).
"""
import ctypes
import importlib.util
import io
import json
import os
import re
import sys
import tempfile
import types

import numpy as np

os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference/scripts/lammps_remcmc.py'

from oracle import oracle as O  # noqa: E402
from neuralmelting_amd import lattice  # noqa: E402


# ---------------------------------------------------------------------------------------------- recording RNG
class Recorder:
    """wraps np.random.rand / randint so that every draw the reference makes is logged in call order"""

    def __init__(self):
        self.log = []
        self._rand, self._randint = np.random.rand, np.random.randint

    def install(self):
        rec = self

        def rand(*a):
            v = rec._rand(*a)
            rec.log.extend(np.atleast_1d(v).ravel().tolist())
            return v

        def randint(*a, **k):
            v = rec._randint(*a, **k)
            rec.log.append(float(v) / 65536.0)  # tags travel as randint/65536 (one uniform format on the tape)
            return v
        np.random.rand, np.random.randint = rand, randint

    def remove(self):
        np.random.rand, np.random.randint = self._rand, self._randint

    def take(self):
        out, self.log = self.log, []
        return out


# ---------------------------------------------------------------------------------------------- fake LAMMPS
class FakeLammps:
    """the 8 methods / 10 command patterns remcmc uses, answered by the oracle"""
    natoms = 256
    sz = 4
    rng = (256, 0, 0)     # (seed, slot, step) of the per-atom Philox streams
    calls = None          # optional list collecting the command strings

    def __init__(self, cmdargs=None):
        self.sim = None

    def file(self, path):
        n = FakeLammps.natoms
        self.sim = O.Sim(n)
        self.sim.set_rng(*FakeLammps.rng)
        box = FakeLammps.sz * lattice.lattice_constant('LJ')
        self.sim.set_box(box)
        self.sim.set_x((lattice.fcc_fractional(FakeLammps.sz) * box).ravel())
        self.sim.set_v(np.zeros(3 * n))
        self.sim.setup()

    def command(self, cmd):
        if FakeLammps.calls is not None:
            FakeLammps.calls.append(cmd)
        s = self.sim
        m = re.fullmatch(r'change_box all x final 0\.0 (\S+) y final 0\.0 (\S+) z final 0\.0 (\S+) units box', cmd)
        if m:
            assert m.group(1) == m.group(2) == m.group(3)
            s.set_box(float(m.group(1)))
            return
        m = re.fullmatch(r'run (\d+)', cmd)
        if m:
            n = int(m.group(1))
            s.setup() if n == 0 else s.run(n)
            return
        m = re.fullmatch(r'displace_atoms all random (\S+) (\S+) (\S+) (\d+) units box', cmd)
        if m:
            assert m.group(1) == m.group(2) == m.group(3)
            s.displace(float(m.group(1)), int(m.group(4)))
            return
        m = re.fullmatch(r'velocity all create (\S+) (\d+) dist gaussian', cmd)
        if m:
            s.velocity_create(float(m.group(1)), int(m.group(2)))
            return
        if cmd == 'velocity all zero linear':
            s.zero_linear()
            return
        if cmd == 'velocity all zero angular':
            s.zero_angular()
            return
        m = re.fullmatch(r'timestep (\S+)', cmd)
        if m:
            s.set_timestep(float(m.group(1)))
            return
        raise NotImplementedError('FakeLammps: ' + cmd)

    def gather_atoms(self, name, typ, cnt):
        a = self.sim.get_x() if name == 'x' else self.sim.get_v()
        return (ctypes.c_double * len(a))(*a)

    def scatter_atoms(self, name, typ, cnt, arr):
        a = np.ctypeslib.as_array(arr).astype(np.float64)
        self.sim.set_x(a) if name == 'x' else self.sim.set_v(a)

    def extract_global(self, name, typ):
        if name == 'natoms':
            return FakeLammps.natoms
        if name == 'boxlo':
            return 0.0
        if name == 'boxhi':
            return self.sim.get_box()
        raise NotImplementedError(name)

    def extract_compute(self, cid, a, b):
        return {'thermo_temp': self.sim.temp, 'thermo_pe': self.sim.pe, 'thermo_ke': self.sim.ke,
                'thermo_press': self.sim.press}[cid]

    def close(self):
        self.sim.close()


def load_reference():
    nb = types.ModuleType('numba')

    def jit(*a, **k):
        if a and callable(a[0]):
            return a[0]
        return lambda f: f
    nb.jit = nb.njit = jit
    sys.modules['numba'] = nb
    lm = types.ModuleType('lammps')
    lm.lammps = FakeLammps
    sys.modules['lammps'] = lm
    orig = np.unravel_index

    def unravel(indices, shape=None, order='C', dims=None):
        return orig(indices, shape if shape is not None else dims, order)
    np.unravel_index = unravel
    spec = importlib.util.spec_from_file_location('lammps_remcmc_ref', REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)   # __main__ block does not run
    return mod


def set_globals(mod, *, el='LJ', sz=4, npn=2, ntn=2, pr=(1.0, 8.0), tr=(0.25, 2.5), mod_=16, ppos=0.125, pvol=0.125,
                nstps=8, bm=True, dx=0.03125, dv=0.03125, nsmpl=4, cutoff=0, name='golden'):
    g = mod.__dict__
    g.update(VERBOSE=0, RESTART=0, PARALLEL=0, DASK=0, DISTRIBUTED=0, INTSTS=0, BM=bm, REFREQ=128, NAME=name, EL=el,
             SZ=sz, NP=npn, NT=ntn, CUTOFF=cutoff, NSMPL=nsmpl, MOD=mod_, PPOS=ppos, PVOL=pvol, NSTPS=nstps, DX=dx, DV=dv,
             SEED=256)
    g['NS'] = npn * ntn
    g['NSWPS'] = nsmpl * mod_
    g['PHMC'] = 1 - ppos - pvol
    g['UNITS'] = {'Ti': 'metal', 'Al': 'metal', 'Ni': 'metal', 'Cu': 'metal', 'LJ': 'lj'}
    g['LAT'] = {'Ti': ('bcc', 2.951), 'Al': ('fcc', 4.046), 'Ni': ('fcc', 3.524), 'Cu': ('fcc', 3.615), 'LJ': ('fcc', 1.122)}
    g['MASS'] = {'Ti': 47.867, 'Al': 29.982, 'Ni': 58.693, 'Cu': 63.546, 'LJ': 1.0}
    g['TIMESTEP'] = {'real': 4.0, 'metal': 0.00390625, 'lj': 0.00390625}
    P32 = np.linspace(pr[0], pr[1], npn, dtype=np.float32)
    T32 = np.linspace(tr[0], tr[1], ntn, dtype=np.float32)
    g['P'] = P32.astype(np.float64)   # float32-rounded values in float64: NumPy-1.x scalar promotion
    g['T'] = T32.astype(np.float64)
    g['DT'] = g['TIMESTEP'][g['UNITS'][el]]
    g['LMPSF'] = 'unused.in'
    return P32, T32


def main():
    mod = load_reference()
    scal = {}

    # ---------------- G1: init_constants (remcmc:114-141)
    g1 = {}
    for tag, kw in (('lj_2x2', dict(npn=2, ntn=2)), ('lj_8x8', dict(npn=8, ntn=8)),
                    ('al_8x8', dict(el='Al', npn=8, ntn=8, pr=(0.0, 8.0), tr=(256.0, 2560.0)))):
        P32, T32 = set_globals(mod, **kw)
        c = mod.init_constants()
        g1[tag] = dict(P=[float(v) for v in P32], T=[float(v) for v in T32], el=kw.get('el', 'LJ'),
                       et=[float(a) for a, b in c], pf=[float(b) for a, b in c])
    scal['G1_constants'] = g1

    # ---------------- G2: gen_mc_param (remcmc:726-745)
    g2 = []
    for ap, av, ah in ((0.0, 0.5, 1.0), (0.49, 0.51, 0.5), (0.25, 0.75, 0.0), (1.0, 0.0, 0.49999)):
        st = [256, None, None, 1.0, -2.0, 3.0, 4.0, 6.1, 226.98] + [0.03125, 0.0625, 0.00390625] + [5, 1, 6, 3, 7, 7] + \
            [np.float32(ap), np.float32(av), np.float32(ah)]
        out = mod.gen_mc_param(st)
        g2.append(dict(ratios=[ap, av, ah], steps_in=st[9:12], steps_out=[float(v) for v in out[9:12]],
                       tail=[float(v) for v in out[12:]]))
    scal['G2_adapt'] = g2

    # ---------------- G3: replica_exchange (remcmc:776-803) on synthetic states
    g3 = []
    rec = Recorder()
    for npn, ntn, seed in ((2, 2, 256), (2, 8, 256), (8, 8, 256), (3, 5, 7)):
        P32, T32 = set_globals(mod, npn=npn, ntn=ntn)
        ns = npn * ntn
        mod.CONST = mod.init_constants()
        rs = np.random.RandomState(1000 + ns)
        # energies / volumes close enough that the sweep produces a mix of accepted and rejected swaps
        pe = -2000.0 + 2.0 * rs.rand(ns) + 0.3 * np.tile(np.arange(ntn), npn)
        ke = 100.0 + rs.rand(ns)
        vol = 225.0 + 0.5 * rs.rand(ns) + 0.05 * np.tile(np.arange(ntn), npn)
        mod.STATE = [[k, None, None, 0.0, float(pe[k]), float(ke[k]), 0.0, 0.0, float(vol[k]), 0.1, 0.2, 0.3] + [0.0] * 9
                     for k in range(ns)]
        np.random.seed(seed)
        rec.install()
        try:
            mod.replica_exchange()
        finally:
            rec.remove()
        g3.append(dict(np=npn, nt=ntn, pe=pe.tolist(), ke=ke.tolist(), vol=vol.tolist(),
                       et=[float(a) for a, b in mod.CONST], pf=[float(b) for a, b in mod.CONST],
                       uniforms=rec.take(), perm=[int(s[0]) for s in mod.STATE]))
    scal['G3_exchange'] = g3

    # ---------------- G4: file formats and command strings (remcmc:176-256, 466, 483, 571, 604, 607)
    P32, T32 = set_globals(mod, npn=2, ntn=2, mod_=128, nsmpl=1024)
    g4 = {}
    with tempfile.TemporaryDirectory() as td:
        cwd = os.getcwd()
        os.chdir(td)
        try:
            out = mod.init_output(3)
            mod.init_header(3, out)
            g4['header_k3'] = open(out[0]).read()
            g4['thrm_name_k3'] = os.path.basename(out[0])
            x = (np.arange(12, dtype=np.float64) * 0.37 - 1.0)
            st = [4, x, x * 0.5, 1.2142857313156128, -1532.6494, 462.21484, 3.9614584, 6.428175, 265.621408,
                  0.03125, 0.029296875, 0.00390625, 12.0, 0.0, 17.0, 9.0, 99.0, 71.0,
                  np.float32(0.0), np.float32(9.0) / np.float32(17.0), np.float32(71.0) / np.float32(99.0)]
            mod.write_thrm(out, st)
            mod.write_traj(out, st)
            g4['thrm_row'] = open(out[0]).read()[len(g4['header_k3']):]
            g4['traj_block'] = open(out[1]).read()
            g4['state'] = dict(natoms=4, x=x.tolist(), box=st[7], row=[float(v) for v in st[3:7]] + [float(st[8])] +
                               [float(v) for v in st[9:21]])
        finally:
            os.chdir(cwd)
    scal['G4_formats'] = g4

    # ---------------- G5: gen_sample traces (remcmc:665-691) with the reference's control flow
    blocks = {}
    cmdlog = {}
    for tag, kw in (('bulk', dict(bm=True, mod_=24, ppos=0.25, pvol=0.25)),
                    ('iter', dict(bm=False, mod_=10, ppos=0.2, pvol=0.3)),
                    ('default_mix', dict(bm=True, mod_=32))):
        P32, T32 = set_globals(mod, npn=2, ntn=2, **kw)
        mod.CONST = mod.init_constants()
        x, v, box, d = lattice.init_states(4, P32, T32, 0.03125, 0.03125)
        FakeLammps.natoms, FakeLammps.sz = 256, 4
        for k in range(4):
            np.random.seed(256 + k)
            FakeLammps.rng = (256, k, 3)
            FakeLammps.calls = [] if k == 0 else None
            state = [256, x[k].copy(), v[k].copy(), 0.0, 0.0, 0.0, 0.0, float(box[k]), float(box[k]) ** 3,
                     float(d[k, 0]), float(d[k, 1]), float(d[k, 2])] + [0.0] * 9
            rec.install()
            try:
                out = mod.gen_sample(k, mod.CONST[k], state)
            finally:
                rec.remove()
            tape = np.array(rec.take())
            if k == 0:
                cmdlog[tag] = FakeLammps.calls[:40]
            pre = '%s_%d_' % (tag, k)
            blocks[pre + 'x_in'], blocks[pre + 'v_in'] = x[k], v[k]
            blocks[pre + 'scal_in'] = np.array([box[k], d[k, 0], d[k, 1], d[k, 2], mod.CONST[k][0], mod.CONST[k][1], mod.T[k % 2]])
            blocks[pre + 'tape'] = tape
            blocks[pre + 'x_out'], blocks[pre + 'v_out'] = np.array(out[1]), np.array(out[2])
            blocks[pre + 'row_out'] = np.array([float(q) for q in out[3:7]] + [float(out[8])] + [float(q) for q in out[9:21]])
            blocks[pre + 'box_out'] = np.array([float(out[7])])
        blocks[tag + '_params'] = np.array([kw['mod_'], kw.get('ppos', 0.125), kw.get('pvol', 0.125), 8, int(kw['bm'])], dtype=np.float64)
    scal['G5_command_strings'] = cmdlog

    with open(os.path.join(HERE, 'ref_scalars.json'), 'w') as f:
        json.dump(scal, f, indent=1)
    np.savez_compressed(os.path.join(HERE, 'ref_blocks.npz'), **blocks)
    print('wrote ref_scalars.json (%d bytes) and ref_blocks.npz (%d bytes)'
          % (os.path.getsize(os.path.join(HERE, 'ref_scalars.json')), os.path.getsize(os.path.join(HERE, 'ref_blocks.npz'))))


if __name__ == '__main__':
    main()
