"""Driver with the reference's command line and file formats (scripts/lammps_remcmc.py, "remcmc") on top of the
MI355X engine.  Same flags (remcmc:22-100), same main loop order (remcmc:959-1001), same `.thrm` / `.traj` /
`.trgt.npy` / `.rstrt.NNNN.npy` outputs, so lammps_parse.py / lammps_distr.py / lammps_vae.py consume a run
unchanged.  What differs by design: gen_samples / gen_mc_params / replica_exchange are one kernel launch each
over all replicas of this rank's pressure rows (include/nm.h) instead of a Dask/joblib fan-out of LAMMPS
instances; the cluster flags (-p -c -d -q -a -nn -np -w -m -nw -nt -mt) are accepted and ignored — ranks come from
torch.distributed.run (one process per GPU).

    python -m neuralmelting_amd.remcmc -v -bm -n remcmc_init -e LJ -ss 4 -pn 8 -tn 8 -sn 1024 -sm 128
"""
import argparse
import os
import sys
import time

import numpy as np

from . import lattice
from .lattice import LAT, MASS, TIMESTEP, UNITS

SEED = 256  # remcmc:851 (hard-coded in the reference, not a flag)


# The command line of the reference (remcmc:22-86), as a table: (short, long, kind, default, what it is here).
# kind: 'flag' = store_true, a type = one value, (type, 2) = two values.  Cluster-control flags are accepted and ignored.
_FLAGS = (
    ('-v', '--verbose', 'flag', None, 'print progress'),
    ('-r', '--restart', 'flag', None, 'start from a restart dump'),
    ('-p', '--parallel', 'flag', None, 'ignored (replicas always run in parallel on the GPU)'),
    ('-c', '--client', 'flag', None, 'ignored (no dask)'),
    ('-d', '--distributed', 'flag', None, 'ignored (ranks come from torch.distributed.run)'),
    ('-is', '--interpolate_states', 'flag', None, 'interpolate the initial volumes along a pressure row, then 1024 NVE steps'),
    ('-bm', '--bulk_move', 'flag', None, 'position moves displace all atoms at once'),
    ('-rd', '--restart_dump', int, 128, 'cycles between restart dumps'),
    ('-rn', '--restart_name', str, 'remcmc_init', 'run name of the dump to restart from'),
    ('-rs', '--restart_step', int, 1024, 'cycle index of the dump to restart from'),
    ('-q', '--queue', str, 'jobqueue', 'ignored'),
    ('-a', '--allocation', str, 'startup', 'ignored'),
    ('-nn', '--nodes', int, 1, 'ignored'),
    ('-np', '--procs_per_node', int, 20, 'ignored'),
    ('-w', '--walltime', int, 72, 'ignored'),
    ('-m', '--memory', int, 32, 'ignored'),
    ('-nw', '--workers', int, 20, 'ignored'),
    ('-nt', '--threads', int, 1, 'ignored'),
    ('-mt', '--method', str, 'fork', 'ignored'),
    ('-n', '--name', str, 'remcmc_init', 'run name (prefix of every file)'),
    ('-e', '--element', str, 'LJ', 'LJ or Al'),
    ('-ss', '--supercell_size', int, 5, 'fcc cells per box edge'),
    ('-pn', '--pressure_number', int, 16, 'points of the pressure grid'),
    ('-pr', '--pressure_range', (float, 2), [1, 8], 'lowest and highest pressure'),
    ('-tn', '--temperature_number', int, 16, 'points of the temperature grid'),
    ('-tr', '--temperature_range', (float, 2), [0.25, 2.5], 'lowest and highest temperature'),
    ('-sc', '--sample_cutoff', int, 0, 'cycles before recording starts'),
    ('-sn', '--sample_number', int, 1024, 'cycles'),
    ('-sm', '--sample_mod', int, 128, 'moves per cycle'),
    ('-pm', '--position_move', float, 0.125, 'probability of a position move'),
    ('-vm', '--volume_move', float, 0.125, 'probability of a volume move'),
    ('-ts', '--timesteps', int, 8, 'velocity-Verlet steps of an HMC move'),
    ('-dx', '--pos_displace', float, 0.03125, 'initial position step (fraction of the lattice parameter)'),
    ('-dv', '--vol_displace', float, 0.03125, 'initial step in log volume'),
)


# Flags the reference does not have (long form only, defaults keep the reference's behaviour).
_EXTRA_FLAGS = (
    ('--revert_rejected_trials', 'flag', None,
     'iterative position moves (no -bm): undo a rejected single-atom trial.  The reference does not (its `od` is a NumPy view '
     'of the coordinates it restores from, remcmc:522-545), so every proposal is kept; that is reproduced by default'),
)


def parse_all(argv=None):
    """(the reference's values in the order of remcmc:89-100, dict of this driver's extra flags)"""
    ap = argparse.ArgumentParser(description='NPT-HMC + replica-exchange sampler on MI355X (flags of lammps_remcmc.py)')
    for short, long_, kind, default, text in _FLAGS:
        if kind == 'flag':
            ap.add_argument(short, long_, action='store_true', help=text)
        elif isinstance(kind, tuple):
            ap.add_argument(short, long_, type=kind[0], nargs=kind[1], default=default, help=text)
        else:
            ap.add_argument(short, long_, type=kind, default=default, help=text)
    for long_, kind, default, text in _EXTRA_FLAGS:
        ap.add_argument(long_, action='store_true', help=text)
    ns = ap.parse_args(argv)
    out = []
    for _, long_, kind, _, _ in _FLAGS:
        val = getattr(ns, long_[2:])
        out.extend(val) if isinstance(kind, tuple) else out.append(val)
    return tuple(out), {long_[2:]: getattr(ns, long_[2:]) for long_, _, _, _ in _EXTRA_FLAGS}


def parse_args(argv=None):
    """the reference's 34 flags with its names and defaults; returns the values in the order of remcmc:89-100
    (two-value ranges flattened)"""
    return parse_all(argv)[0]


def init_constant(P, T, el, i, j):
    """(et, pf) of one replica, remcmc:114-132, in float64 on the float32-rounded grid values (the NumPy-1.x
    promotion the reference was written for)"""
    p, t = float(P[i]), float(T[j])
    if UNITS[el] == 'real':
        na = 6.0221409e23
        kb = 3.29983e-27
        r = kb * na
        return r * t, 1e-30 * (1.01325e5 * p) / (4.184e3 * kb * t)
    if UNITS[el] == 'metal':
        kb = 8.61733e-5
        return kb * t, 1e-30 * (1e5 * p) / (1.60218e-19 * kb * t)
    kb = 1.0
    return kb * t, p / (kb * t)


class Run:
    """one invocation of the driver: holds what the reference keeps in module globals"""

    def __init__(self, argv=None, cwd=None, rank=0, world=1, device=0):
        (self.VERBOSE, self.RESTART, _par, _dask, _dist, self.INTSTS, self.BM, self.REFREQ, self.RENAME, self.RESTEP,
         _q, _a, _nn, _ppn, _w, _m, _nw, _nth, _mt, self.NAME, self.EL, self.SZ, self.NP, self.LP, self.HP,
         self.NT, self.LT, self.HT, self.CUTOFF, self.NSMPL, self.MOD, self.PPOS, self.PVOL, self.NSTPS,
         self.DX, self.DV), extra = parse_all(argv)
        self.ITER_REVERT = bool(extra['revert_rejected_trials'])
        self.cwd = os.getcwd() if cwd is None else cwd
        self.rank, self.world, self.device = rank, world, device
        self.NS = self.NP * self.NT
        self.NSWPS = self.NSMPL * self.MOD
        self.PHMC = 1 - self.PPOS - self.PVOL
        self.P = np.linspace(self.LP, self.HP, self.NP, dtype=np.float32)  # remcmc:895
        self.T = np.linspace(self.LT, self.HT, self.NT, dtype=np.float32)  # remcmc:897
        self.DT = TIMESTEP[UNITS[self.EL]]
        self.PREF = self.cwd + '/%s.%s.%s.lammps' % (self.NAME, self.EL.lower(), LAT[self.EL][0])
        if LAT[self.EL][0] != 'fcc':
            raise NotImplementedError('only fcc elements are supported')
        self.natoms = 4 * self.SZ ** 3
        # contiguous pressure rows per rank: the exchange never leaves a row (remcmc:782-798), so it stays on the device.
        # With fewer rows than ranks the slots are dealt out evenly instead and the sweep runs over RCCL (exchange.py).
        self.split_rows = world > 1 and self.NP < world and self.NS % world == 0
        if os.environ.get('NM_FORCE_SPLIT_ROWS') == '1' and self.NS % world == 0:
            self.split_rows = True   # the host-side (collective) exchange also where whole rows would do: lets ONE rank run it
        self._etpf = None
        if self.split_rows:
            self.nloc = self.NS // world
            self.k0 = rank * self.nloc
            self.row0, self.nrows = self.k0 // self.NT, 0
        else:
            base, extra = divmod(self.NP, world)
            self.nrows = base + (1 if rank < extra else 0)
            self.row0 = rank * base + min(rank, extra)
            self.k0 = self.row0 * self.NT
            self.nloc = self.nrows * self.NT
        self.engine = None
        self.STEP = -1

    # ------------------------------------------------------------------ outputs (remcmc:148-316)
    def file_prefix(self, i, j):
        return self.cwd + '/%s.%s.%s.%02d.%02d.lammps' % (self.NAME, self.EL.lower(), LAT[self.EL][0], i, j)

    def init_output(self, k):
        i, j = divmod(k, self.NT)
        thrm = self.file_prefix(i, j) + '.thrm'
        traj = thrm.replace('thrm', 'traj')
        for f in (thrm, traj):  # the reference means to remove both (its second test repeats `thrm`, remcmc:161-164)
            if os.path.isfile(f):
                os.remove(f)
        return thrm, traj

    def header_text(self, k):
        """remcmc:176-209"""
        i, j = divmod(k, self.NT)
        el = self.EL
        lines = ['# ---------------------', '# simulation parameters', '# ---------------------',
                 '# nsmpl:    %d' % self.NSMPL, '# cutoff:   %d' % self.CUTOFF, '# mod:      %d' % self.MOD,
                 '# nswps:    %d' % self.NSWPS, '# ppos:     %f' % self.PPOS, '# pvol:     %f' % self.PVOL,
                 '# phmc:     %f' % self.PHMC, '# nstps:    %d' % self.NSTPS, '# seed:     %d' % SEED,
                 '# ---------------------', '# material properties', '# ---------------------',
                 '# element:  %s' % el, '# units:    %s' % UNITS[el], '# lattice:  %s' % LAT[el][0],
                 '# latpar:   %f' % LAT[el][1], '# size:     %d' % self.SZ, '# mass:     %f' % MASS[el],
                 '# press:    %f' % self.P[i], '# temp:     %f' % self.T[j], '# dx:       %f' % self.DX,
                 '# dv:       %f' % self.DV, '# dt:       %f' % self.DT,
                 '# -----------------------------------------------------------------------------------------------',
                 '# | tmp | pe | ke | vir | vol | dx | dv | dt | ntp | nap | ntv | nav | nth | nah | ap | av | ah |',
                 '# -----------------------------------------------------------------------------------------------']
        return '\n'.join(lines) + '\n'

    @staticmethod
    def thrm_text(row):
        """one .thrm row: 17 x ' %.4E' (remcmc:235-245), formatted by the library's host-side formatter"""
        import ctypes as C
        from . import _lib as B
        r = np.ascontiguousarray(row, dtype=np.float64)
        buf = C.create_string_buffer(17 * 14 + 8)
        n = B.load().nm_format_thrm(r.ctypes.data_as(B.c_double_p), buf, len(buf))
        if n < 0:
            raise RuntimeError('nm_format_thrm failed')
        return buf.raw[:n].decode()

    @staticmethod
    def traj_text(natoms, box, x):
        """one .traj frame (remcmc:248-256): 'natoms box' then natoms lines of 3 x ' %.4E'"""
        import ctypes as C
        from . import _lib as B
        xx = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
        buf = C.create_string_buffer(64 + 42 * int(natoms))
        n = B.load().nm_format_traj(int(natoms), float(box), xx.ctypes.data_as(B.c_double_p), buf, len(buf))
        if n < 0:
            raise RuntimeError('nm_format_traj failed')
        return buf.raw[:n].decode()

    def init_outputs(self):
        self.OUTPUT = {k: self.init_output(k) for k in range(self.k0, self.k0 + self.nloc)}

    def init_headers(self):
        for k, out in self.OUTPUT.items():
            with open(out[0], 'w') as f:
                f.write(self.header_text(k))

    def write_outputs(self, rows, x, box):
        """write_outputs (remcmc:265-286) for this rank's replicas: one C call formats and appends all files (threaded)"""
        import ctypes as C
        from . import _lib as B
        ks = range(self.k0, self.k0 + self.nloc)
        thrm = (C.c_char_p * self.nloc)(*[self.OUTPUT[k][0].encode() for k in ks])
        traj = (C.c_char_p * self.nloc)(*[self.OUTPUT[k][1].encode() for k in ks])
        rows = np.ascontiguousarray(rows, dtype=np.float64)
        x = np.ascontiguousarray(x, dtype=np.float64)
        box = np.ascontiguousarray(box, dtype=np.float64)
        rc = B.load().nm_append_outputs(self.nloc, self.natoms, thrm, traj, rows.ctypes.data_as(B.c_double_p),
                                        x.ctypes.data_as(B.c_double_p), box.ctypes.data_as(B.c_double_p), 0)
        if rc != 0:
            raise IOError('nm_append_outputs failed (%d)' % rc)

    def _write_async(self, rows, x, box):
        """write a recorded cycle on a helper thread (the C call releases the GIL) while the host drives the next block;
        cycles are written in order: the previous write is joined first"""
        import threading
        self._write_join()
        self._writer_err = None

        def work():
            try:
                self.write_outputs(rows, x, box)
            except BaseException as e:  # surfaced by the next join
                self._writer_err = e
        self._writer = threading.Thread(target=work)
        self._writer.start()

    def _write_join(self):
        w = getattr(self, '_writer', None)
        if w is not None:
            w.join()
            self._writer = None
            if self._writer_err is not None:
                raise self._writer_err

    def consolidate_outputs(self):
        """remcmc:289-316 (rank 0, after every rank finished writing)"""
        thrm = [self.file_prefix(*divmod(k, self.NT)) + '.thrm' for k in range(self.NS)]
        traj = [t.replace('thrm', 'traj') for t in thrm]
        import shutil
        for ext, files in (('.thrm', thrm), ('.traj', traj)):
            with open(self.PREF + ext, 'wb') as out:     # same bytes as the reference's line-by-line copy
                for k in range(self.NS):
                    with open(files[k], 'rb') as fin:
                        shutil.copyfileobj(fin, out, 1 << 22)
        for k in range(self.NS):
            os.remove(thrm[k])
            os.remove(traj[k])

    # ------------------------------------------------------------------ restart (remcmc:810-828)
    def state_lists(self):
        """the reference's STATE: one 21-entry list per local replica (remcmc:432-433, 690-691)"""
        x, v, box, d = self.engine.get_state()
        rows = self.engine.thermo()
        out = []
        for q in range(self.nloc):
            r = rows[q]
            out.append([self.natoms, x[q].copy(), v[q].copy(), r[0], r[1], r[2], r[3], box[q], r[4], d[q, 0], d[q, 1], d[q, 2],
                        r[8], r[9], r[10], r[11], r[12], r[13], np.float32(r[14]), np.float32(r[15]), np.float32(r[16])])
        return out

    def restart_file(self, name, step):
        return self.cwd + '/%s.%s.%s.lammps.rstrt.%04d.npy' % (name, self.EL.lower(), LAT[self.EL][0], step)

    def dump_samples_restart(self):
        """np.save of the object array NS x 21 (remcmc:821-828); ranks gather through files on the node"""
        state = self.state_lists()
        rf = self.restart_file(self.NAME, self.STEP + 1)
        if self.world == 1:
            np.save(rf, np.array(state, dtype=object))
            return
        part = rf + '.part%03d.npy' % self.rank
        np.save(part, np.array(state, dtype=object))
        self.barrier()
        if self.rank == 0:
            full = []
            for r in range(self.world if self.split_rows else min(self.world, self.NP)):  # ranks without rows hold nothing
                p = rf + '.part%03d.npy' % r
                full.extend(list(np.load(p, allow_pickle=True)))
                os.remove(p)
            np.save(rf, np.array(full, dtype=object))
        self.barrier()

    def load_samples_restart(self):
        """remcmc:810-818 (allow_pickle is required on current NumPy)"""
        rf = self.restart_file(self.RENAME, self.RESTEP)
        state = list(np.load(rf, allow_pickle=True))[self.k0:self.k0 + self.nloc]
        x = np.array([np.asarray(s[1], dtype=np.float64) for s in state])
        v = np.array([np.asarray(s[2], dtype=np.float64) for s in state])
        box = np.array([float(s[7]) for s in state])
        d = np.array([[float(s[9]), float(s[10]), float(s[11])] for s in state])
        th = np.array([[float(s[3]), float(s[4]), float(s[5]), float(s[6]), float(s[8])] for s in state])
        return x, v, box, d, th

    # ------------------------------------------------------------------ plumbing
    def barrier(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()

    def log(self, *a):
        if self.VERBOSE and self.rank == 0:
            print(*a, flush=True)

    def make_engine(self):
        from .engine import Engine
        kw = dict(slot0=self.k0, nslots=self.nloc) if self.split_rows else dict(row0=self.row0, nrows=self.nrows)
        return Engine(self.natoms, self.P, self.T, element=self.EL, ppos=self.PPOS, pvol=self.PVOL, nstps=self.NSTPS,
                      bulk=self.BM, seed=SEED, device=self.device, iter_revert=self.ITER_REVERT, **kw)

    def _quiet_cycles(self):
        """how many cycles from STEP on neither record (remcmc:983-985), nor dump the restart file (remcmc:990-992), nor are the run's last
        (remcmc:994-995: no exchange behind it) — those can go to the engine as one call (Engine.run_cycles); 0 where a pressure row
        is split across ranks (its exchange needs the host)"""
        if getattr(self, 'split_rows', False) or not hasattr(self.engine, 'run_cycles'):
            return 0
        n, s = 0, self.STEP
        while (s + 1) <= self.CUTOFF and (s + 1) % self.REFREQ != 0 and (s + 1) != self.NSMPL:
            n += 1
            s += 1
        return n

    def replica_exchange(self, step):
        """replica_exchange (remcmc:776-803): on the device when this rank owns whole rows, else all-gather + identical sweep"""
        eng = self.engine
        if not self.split_rows:
            eng.set_step(step)
            return eng.exchange(count=bool(self.VERBOSE))
        from . import exchange as X
        import torch.distributed as dist
        # gloo (CPU rehearsals, also on a GPU box) moves host tensors; a single forced rank has no group of its own
        if dist.is_initialized():
            info = (dist.get_world_size(), dist.get_backend() == 'nccl')
        else:
            info = (1, False)
        if self._etpf is None:
            self._etpf = (np.array([init_constant(self.P, self.T, self.EL, *divmod(k, self.NT))[0] for k in range(self.NS)]),
                          np.array([init_constant(self.P, self.T, self.EL, *divmod(k, self.NT))[1] for k in range(self.NS)]))
        return X.exchange_split(eng, step, self.NP, self.NT, SEED, self.k0, self.natoms, self._etpf[0], self._etpf[1], info,
                                rank=self.rank)

    # ------------------------------------------------------------------ main (remcmc:834-1001)
    def main(self):
        np.random.seed(SEED)
        if self.rank == 0:
            np.save(self.PREF + '.virial.trgt.npy', self.P)  # remcmc:903-904
            np.save(self.PREF + '.temp.trgt.npy', self.T)
        if self.nloc == 0:  # more ranks than pressure rows: this rank only keeps the collectives company
            return self._idle()
        self.engine = self.make_engine()
        eng = self.engine
        if not self.BM and not self.ITER_REVERT:
            self.log('note: iterative position moves follow the reference literally: a rejected single-atom trial is counted '
                     'but not undone (remcmc:522-545); --revert_rejected_trials gives the corrected move')
        self.init_outputs()
        if self.CUTOFF < self.NSMPL:
            self.init_headers()
        if self.RESTART:
            self.log('loading samples from previous dump')
            x, v, box, d, th = self.load_samples_restart()
            eng.set_state(x, v, box, d)
            eng.set_thermo(th)
            n = self.replica_exchange(0xFFFFFFFF)  # the exchange after a restart draws from its own counter block
            if self.VERBOSE:
                self.log('%d replica exchanges performed' % n)
        else:
            self.log('initializing samples')
            r0 = self.k0 // self.NT                                    # rows covering this rank's slots
            r1 = (self.k0 + self.nloc - 1) // self.NT
            x, v, box, d = lattice.init_states(self.SZ, self.P, self.T, self.DX, self.DV, el=self.EL, seed=SEED,
                                               row0=r0, nrows=r1 - r0 + 1, interpolate=self.INTSTS)
            a = self.k0 - r0 * self.NT
            x, v, box, d = x[a:a + self.nloc], v[a:a + self.nloc], box[a:a + self.nloc], d[a:a + self.nloc]
            eng.set_state(x, v, box, d)
            if self.INTSTS:
                eng.set_step(0xFFFFFFFE)
                eng.run_md(1024)  # velocity create / zero / run 1024 of the -is branch (remcmc:421-425)
            else:
                eng.run_block(0)  # "run 0" of init_sample: thermo scalars of the initial states (remcmc:427)
        self.STEP = -1
        self.dump_samples_restart()
        snaps = 0  # recorded cycles whose outputs are on their way to the host (Engine.snapshot): fetched and written one cycle later, while the next
        # block runs on the GPU — the stream never waits for the copies, the formatting or the files
        eng.synchronize()
        t_loop = time.perf_counter()
        self.STEP = 0
        while self.STEP < self.NSMPL:
            eng.set_step(self.STEP)
            quiet = self._quiet_cycles()
            if quiet > 1:                                 # cycles that write nothing and dump nothing: one call, one launch where the
                eng.run_cycles(quiet, self.MOD)           # grid has the kernel for it (nm_run_cycles) — the same chains
                if self.VERBOSE:
                    self.log('cycles %d-%d: outputs off' % (self.STEP, self.STEP + quiet - 1))
                self.STEP += quiet
                continue
            eng.run_block(self.MOD)                       # gen_samples (asynchronous)
            record = (self.STEP + 1) > self.CUTOFF        # remcmc:983-985
            if record:
                eng.snapshot()                            # the 17 thermo columns, positions and box as of here: before gen_mc_params zeroes the counters
                snaps += 1
            eng.adapt()                                   # gen_mc_params
            if (self.STEP + 1) % self.REFREQ == 0:
                self.dump_samples_restart()               # remcmc:990-992
            if (self.STEP + 1) != self.NSMPL:             # remcmc:994-995
                n = self.replica_exchange(self.STEP)
                if self.VERBOSE:
                    self.log('%d replica exchanges performed' % n)
            while snaps > (1 if record else 0):           # the cycle before this one: write_outputs (remcmc:259-286), in cycle order
                self._write_async(*eng.snapshot_fetch())
                snaps -= 1
            self.STEP += 1
        self.STEP = self.NSMPL - 1
        while snaps:
            self._write_async(*eng.snapshot_fetch())
            snaps -= 1
        self._write_join()
        eng.synchronize()
        self.loop_seconds = time.perf_counter() - t_loop   # the metric's clock: main loop, remcmc:977-995
        if self.VERBOSE:
            self.log('main loop: %.3f s, %.0f MC sweeps/s on this rank (%d replicas x %d moves x %d cycles)'
                     % (self.loop_seconds, self.nloc * self.MOD * self.NSMPL / max(self.loop_seconds, 1e-9), self.nloc, self.MOD,
                        self.NSMPL))
        self.barrier()
        if self.CUTOFF < self.NSMPL and self.rank == 0:
            self.consolidate_outputs()
        self.barrier()
        eng.close()

    def _idle(self):
        self.barrier()          # dump at STEP = -1
        self.barrier()
        for step in range(self.NSMPL):
            if (step + 1) % self.REFREQ == 0:
                self.barrier()
                self.barrier()
        self.barrier()
        self.barrier()


def main(argv=None):
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 or 'RANK' in os.environ:   # under torch.distributed.run: one rank per GPU over RCCL (also a world of one)
        import torch
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        use_gpu = torch.cuda.is_available()
        # one rank per GPU over RCCL.  NM_DIST_BACKEND=gloo: rehearsal with more ranks than cards (RCCL refuses two ranks on one
        # device); the ranks then share the cards round-robin and the few host-side collectives go over gloo.
        backend = os.environ.get('NM_DIST_BACKEND', 'nccl' if use_gpu else 'gloo')
        if use_gpu:
            ndev = torch.cuda.device_count()
            if backend == 'gloo':
                local = local % ndev   # rehearsal: the ranks share the cards round-robin
            elif local >= ndev:
                raise SystemExit('rank %d: LOCAL_RANK %d but only %d GPU(s) visible; RCCL needs one device per rank '
                                 '(NM_DIST_BACKEND=gloo rehearses more ranks than cards)' % (rank, local, ndev))
            torch.cuda.set_device(local)
        dist.init_process_group(backend)
    run = Run(argv, rank=rank, world=world, device=local)
    run_guarded(run)


def run_guarded(run):
    """run.main() of one rank.  An error on one rank of several must not leave the others waiting in a barrier or a collective for
    ever: the failing rank leaves at once with a failure code and the launcher (torch.distributed.run) ends the rest."""
    try:
        if os.environ.get('NM_TESTING') == '1' and os.environ.get('NM_TEST_FAIL_RANK') == str(run.rank):   # test hook: tests/test_multiproc.py
            raise RuntimeError('injected failure on rank %d' % run.rank)
        run.main()
    except BaseException:
        if run.world > 1:
            import traceback
            traceback.print_exc()
            sys.stderr.flush()
            os._exit(1)
        raise
    dist = sys.modules.get('torch.distributed')   # only a process group that was created is destroyed: a single-process run never
    if dist is not None and dist.is_available() and dist.is_initialized():   # imports torch (and must not need it to exit cleanly)
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
