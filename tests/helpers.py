"""shared helpers of the parity tests: the reference's main loop (remcmc:977-995) driven on the oracle"""
import numpy as np

from neuralmelting_amd import lattice

LAT_LJ = 1.122


def grids(npn, ntn, pr=(1.0, 8.0), tr=(0.25, 2.5)):
    """P, T float32 grids, remcmc:895-897"""
    return np.linspace(pr[0], pr[1], npn, dtype=np.float32), np.linspace(tr[0], tr[1], ntn, dtype=np.float32)


def constants_lj(P, T, row0=0, nrows=None):
    """init_constant for lj units in float64 on the float32 grid values (remcmc:128-131)"""
    nrows = len(P) - row0 if nrows is None else nrows
    et = np.array([float(T[j]) for r in range(nrows) for j in range(len(T))])
    pf = np.array([float(P[row0 + r]) / float(T[j]) for r in range(nrows) for j in range(len(T))])
    tq = et.copy()
    return et, pf, tq


def constants_metal(P, T, row0=0, nrows=None):
    """init_constant for metal units (remcmc:124-127)"""
    nrows = len(P) - row0 if nrows is None else nrows
    kb = 8.61733e-5
    et = np.array([kb * float(T[j]) for r in range(nrows) for j in range(len(T))])
    pf = np.array([1e-30 * (1e5 * float(P[row0 + r])) / (1.60218e-19 * kb * float(T[j])) for r in range(nrows) for j in range(len(T))])
    tq = np.array([float(T[j]) for r in range(nrows) for j in range(len(T))])
    return et, pf, tq


class OracleLoop:
    """gen_samples / gen_mc_params / replica_exchange on the oracle, holding STATE like the reference does"""

    def __init__(self, O, sz, P, T, *, dx=0.03125, dv=0.03125, ppos=0.125, pvol=0.125, nstps=8, bulk=True, seed=256,
                 row0=0, nrows=None, iter_revert=False, nthreads=0, el='LJ', k0=None, nk=None):
        self.O = O
        self.P, self.T = P, T
        self.nt = len(T)
        self.row0 = row0
        self.nrows = len(P) - row0 if nrows is None else nrows
        self.ns = self.nrows * self.nt
        self.natoms = 4 * sz ** 3
        self.el = el
        self.kw = dict(natoms=self.natoms, nstps=nstps, bulk=bulk, ppos=ppos, pvol=pvol, lat=lattice.LAT[el][1], seed=seed,
                       iter_revert=iter_revert, nthreads=nthreads)
        if el == 'Al':
            self.kw.update(units=1, mass=lattice.MASS['Al'], pot=1)
        self.seed = seed
        self.x, self.v, self.box, self.d = lattice.init_states(sz, P, T, dx, dv, el=el, seed=seed, row0=row0, nrows=self.nrows)
        self.et, self.pf, self.tq = (constants_lj if el == 'LJ' else constants_metal)(P, T, row0, self.nrows)
        self.slot0 = row0 * self.nt
        if nk is not None:  # an arbitrary slot range inside the covering rows (a pressure row split across ranks)
            a = k0 - row0 * self.nt
            sl = slice(a, a + nk)
            self.x, self.v, self.box, self.d = self.x[sl], self.v[sl], self.box[sl], self.d[sl]
            self.et, self.pf, self.tq = self.et[sl], self.pf[sl], self.tq[sl]
            self.ns, self.slot0 = nk, k0
        self.thermo = np.zeros((self.ns, 5))
        self.thermo[:, 4] = self.box ** 3
        self.counters = np.zeros((self.ns, 6))
        self.ratios = np.zeros((self.ns, 3), dtype=np.float32)

    def run_block(self, mod, step):
        out = self.O.run_blocks(self.x, self.v, self.box, self.d, self.tq, self.et, self.pf, mod=mod,
                                slot0=self.slot0, step=step, **self.kw)
        self.x, self.v, self.box = out['x'], out['v'], out['box']
        self.thermo, self.counters, self.ratios = out['thermo'], out['counters'], out['ratios']

    def rows(self):
        """the 17 .thrm columns (remcmc:235-245)"""
        return np.concatenate([self.thermo[:, :5], self.d, self.counters, self.ratios.astype(np.float64)], axis=1)

    def adapt(self):
        for k in range(self.ns):
            self.d[k] = self.O.adapt(self.ratios[k], self.d[k])
        self.counters[:] = 0
        self.ratios[:] = 0

    def exchange(self, step, tape=None):
        etot = self.thermo[:, 1] + self.thermo[:, 2]
        swaps, perm, _, _, crit = self.O.exchange(len(self.P), self.nt, self.row0, self.nrows, self.seed, step, etot,
                                                  self.thermo[:, 4], self.et, self.pf, tape=tape)
        # entries [0..11] travel: configuration, thermo scalars, dx dv dt (remcmc:798)
        self.x, self.v, self.box = self.x[perm], self.v[perm], self.box[perm]
        self.thermo, self.d = self.thermo[perm], self.d[perm]
        return swaps, perm, crit


class OracleEngine:
    """TEST-ONLY stand-in with the Engine interface, backed by the oracle, so that the driver's host logic and file
    formats can be exercised on a machine without a GPU (never used by the product: Run.make_engine builds the HIP one)"""

    def __init__(self, O, run):
        self.O, self.run = O, run
        sz = run.SZ
        kw = {}
        if getattr(run, 'split_rows', False):
            r0 = run.k0 // run.NT
            r1 = (run.k0 + run.nloc - 1) // run.NT
            kw = dict(row0=r0, nrows=r1 - r0 + 1, k0=run.k0, nk=run.nloc)
        else:
            kw = dict(row0=run.row0, nrows=run.nrows)
        self.loop = OracleLoop(O, sz, run.P, run.T, dx=run.DX, dv=run.DV, ppos=run.PPOS, pvol=run.PVOL, nstps=run.NSTPS,
                               bulk=run.BM, **kw)
        self.nslots = self.loop.ns
        self.natoms = self.loop.natoms
        self.step = 0

    def set_state(self, x=None, v=None, box=None, dxdvdt=None, k0=0, nk=None):
        lp = self.loop
        nk = lp.ns - k0 if nk is None else nk
        sl = slice(k0, k0 + nk)
        if x is not None: lp.x[sl] = np.array(x, dtype=np.float64).reshape(nk, -1)
        if v is not None: lp.v[sl] = np.array(v, dtype=np.float64).reshape(nk, -1)
        if box is not None:
            lp.box[sl] = np.array(box, dtype=np.float64).reshape(nk)
            lp.thermo[sl, 4] = lp.box[sl] ** 3
        if dxdvdt is not None: lp.d[sl] = np.array(dxdvdt, dtype=np.float64).reshape(nk, 3)

    def set_thermo(self, th, k0=0, nk=None):
        nk = self.loop.ns - k0 if nk is None else nk
        self.loop.thermo[k0:k0 + nk] = np.array(th, dtype=np.float64).reshape(nk, 5)

    def get_state(self, k0=0, nk=None, velocities=True):
        lp = self.loop
        nk = lp.ns - k0 if nk is None else nk
        sl = slice(k0, k0 + nk)
        return lp.x[sl].copy(), lp.v[sl].copy(), lp.box[sl].copy(), lp.d[sl].copy()

    def get_slots(self, slots, velocities=True):
        lp, sl = self.loop, np.asarray(slots, dtype=int)
        return lp.x[sl].copy(), lp.v[sl].copy(), lp.box[sl].copy(), lp.d[sl].copy(), lp.thermo[sl].copy()

    def set_slots(self, slots, x=None, v=None, box=None, dxdvdt=None, th=None):
        lp, sl = self.loop, np.asarray(slots, dtype=int)
        nk = len(sl)
        if x is not None: lp.x[sl] = np.array(x, dtype=np.float64).reshape(nk, -1)
        if v is not None: lp.v[sl] = np.array(v, dtype=np.float64).reshape(nk, -1)
        if box is not None:
            lp.box[sl] = np.array(box, dtype=np.float64).reshape(nk)
            lp.thermo[sl, 4] = lp.box[sl] ** 3
        if dxdvdt is not None: lp.d[sl] = np.array(dxdvdt, dtype=np.float64).reshape(nk, 3)
        if th is not None: lp.thermo[sl] = np.array(th, dtype=np.float64).reshape(nk, 5)

    def snapshot(self):
        if not hasattr(self, '_snaps'):
            self._snaps = []
        assert len(self._snaps) < 2, 'two snapshots are pending'
        lp = self.loop
        self._snaps.append((lp.rows().copy(), lp.x.copy(), lp.box.copy()))

    def snapshot_fetch(self, positions=True):
        rows, x, box = self._snaps.pop(0)
        return rows, (x if positions else None), box

    def set_step(self, step): self.step = int(step)
    def run_block(self, mod): self.loop.run_block(mod, self.step)

    def run_cycles(self, ncycles, mod):
        for k in range(ncycles):
            self.loop.run_block(mod, self.step + k); self.loop.adapt(); self.loop.exchange(self.step + k)
    def thermo(self): return self.loop.rows()
    def adapt(self): self.loop.adapt()

    def exchange(self, count=True):
        return self.loop.exchange(self.step)[0]

    def run_md(self, nsteps):
        lp = self.loop
        for k in range(lp.ns):
            s = self.O.Sim(lp.natoms)
            s.set_rng(lp.seed, lp.slot0 + k, self.step)
            s.set_box(self.O.q6(lp.box[k])); s.set_x(lp.x[k]); s.set_v(lp.v[k]); s.setup()
            s.velocity_create(self.O.q6(lp.tq[k]), 0); s.zero_linear(); s.zero_angular()
            s.set_timestep(self.O.q6(lp.d[k, 2])); s.setup(); s.run(nsteps)
            lp.x[k], lp.v[k], lp.box[k] = s.get_x(), s.get_v(), s.get_box()
            lp.thermo[k] = [s.temp, s.pe, s.ke, s.press, s.get_box() ** 3]

    def synchronize(self): pass
    def close(self): pass


class BenchOracleEngine(OracleEngine):
    """TEST-ONLY stand-in with the constructor and the measurement methods of neuralmelting_amd.Engine, backed by the oracle: lets
    bench.py's rank / leg / reduction logic run on CPUs over gloo (tests/_mp_bench.py hands it to bench.main(make_engine=...));
    bench.py itself never builds one."""

    def __init__(self, O, natoms, P, T, *, element='LJ', ppos=0.125, pvol=0.125, nstps=8, bulk=True, seed=256, device=0, row0=0,
                 nrows=None, iter_revert=False, slot0=None, nslots=None):
        import time
        self._time = time
        self.O = O
        sz = 1
        while 4 * sz ** 3 < natoms:
            sz += 1
        nt = len(T)
        kw = dict(row0=row0, nrows=nrows)
        if nslots is not None:
            r0, r1 = slot0 // nt, (slot0 + nslots - 1) // nt
            kw = dict(row0=r0, nrows=r1 - r0 + 1, k0=slot0, nk=nslots)
        self.loop = OracleLoop(O, sz, P, T, ppos=ppos, pvol=pvol, nstps=nstps, bulk=bulk, seed=seed, el=element, **kw)
        self.nslots, self.natoms, self.step = self.loop.ns, self.loop.natoms, 0
        self.cus_per_replica, self.heals = 1, 0
        self._launches, self._ms, self._blocks = 0, 0.0, 0

    def run_block(self, mod):
        t0 = self._time.perf_counter()
        OracleEngine.run_block(self, mod)
        self._ms += (self._time.perf_counter() - t0) * 1e3
        self._launches += 1
        self._blocks += 1

    def run_cycles(self, ncycles, mod):
        for k in range(ncycles):
            self.run_block(mod); self.adapt(); self.exchange(count=False)
            self.step += 1
        self.step -= ncycles

    def constants(self): return self.loop.et.copy(), self.loop.pf.copy()
    def timing_reset(self): self._launches, self._ms = 0, 0.0
    def timing(self): return self._launches, self._ms

    def stats(self, reset=False):
        s = np.zeros((self.nslots, 10))
        s[:, 0] = s[:, 2] = 7.0 * self._blocks      # evaluations (nominal), energy evaluations
        s[:, 3] = 1000.0 * s[:, 2]                  # interacting pairs summed over those
        s[:, 4] = 1.0
        s[:, 6] = max(self._blocks, 1)
        if reset:
            self._blocks = 0
        return s
