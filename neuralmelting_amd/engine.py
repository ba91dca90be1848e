"""Engine — Python face of the C-ABI (include/nm.h): one context per GPU holding a contiguous range of
pressure rows of the P x T replica grid.  Mirrors the per-replica functions the reference's orchestration
maps over: gen_samples -> run_block, gen_mc_params -> adapt, replica_exchange -> exchange."""
import ctypes as C

import numpy as np

from . import _lib as B


class NMError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, 'nm error %d: %s' % (code, msg))
        self.code = code


def _dp(a):
    return None if a is None else a.ctypes.data_as(B.c_double_p)


class Engine:
    """batched replica engine on one MI355X"""

    ELEMENTS = {'LJ': B.NM_EL_LJ, 'Al': B.NM_EL_AL}

    def __init__(self, natoms, P, T, *, element='LJ', ppos=0.125, pvol=0.125, nstps=8, bulk=True, seed=256,
                 device=0, row0=0, nrows=None, iter_revert=False, slot0=None, nslots=None):
        self.lib = B.load()
        self._P = np.ascontiguousarray(P, dtype=np.float32)
        self._T = np.ascontiguousarray(T, dtype=np.float32)
        self.np, self.nt = len(self._P), len(self._T)
        self.row0 = int(row0)
        self.nrows = self.np - self.row0 if nrows is None else int(nrows)
        cfg = B.NMConfig()
        cfg.size = C.sizeof(B.NMConfig)
        cfg.element = self.ELEMENTS[element]
        cfg.natoms = int(natoms)
        cfg.np, cfg.nt, cfg.row0, cfg.nrows = self.np, self.nt, self.row0, self.nrows
        cfg.nstps, cfg.bulk, cfg.iter_revert = int(nstps), int(bool(bulk)), int(bool(iter_revert))
        cfg.device, cfg.seed = int(device), int(seed)
        cfg.ppos, cfg.pvol = float(ppos), float(pvol)
        if nslots is not None:  # an arbitrary global slot range (a pressure row split across GPUs)
            cfg.slot0, cfg.nslots = int(slot0 or 0), int(nslots)
        cfg.P = self._P.ctypes.data_as(B.c_float_p)
        cfg.T = self._T.ctypes.data_as(B.c_float_p)
        h = C.c_void_p()
        rc = self.lib.nm_create(C.byref(cfg), C.byref(h))
        if rc != B.NM_OK:
            raise NMError(rc, self.lib.nm_last_error(None).decode())
        self.h = h
        self.natoms = int(natoms)
        self.nslots = self.lib.nm_nslots(self.h)
        self.cus_per_replica = self.lib.nm_cus_per_replica(self.h)

    # -- plumbing
    def _chk(self, rc):
        if rc != B.NM_OK:
            raise NMError(rc, self.lib.nm_last_error(self.h).decode())

    def _settled(self, rc):
        """after a call that looks at the queue's outcome (and may have re-issued a block with fewer workgroups per replica)"""
        self._chk(rc)
        self.cus_per_replica = self.lib.nm_cus_per_replica(self.h)

    @property
    def heals(self):
        """blocks re-issued at fewer workgroups per replica so far (nm_heal_count)"""
        return self.lib.nm_heal_count(self.h)

    def close(self):
        if getattr(self, 'h', None):
            self.lib.nm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- constants / state
    def constants(self):
        """(et, pf) per local slot: init_constants, remcmc:114-141"""
        et, pf = np.empty(self.nslots), np.empty(self.nslots)
        self._chk(self.lib.nm_get_const(self.h, _dp(et), _dp(pf)))
        return et, pf

    def set_state(self, x=None, v=None, box=None, dxdvdt=None, k0=0, nk=None):
        nk = self.nslots - k0 if nk is None else nk
        n3 = 3 * self.natoms

        def prep(a, shape):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=np.float64)
            if a.size != int(np.prod(shape)):
                raise ValueError('bad array size %d, expected %s' % (a.size, shape))
            return a
        x, v, box, dxdvdt = prep(x, (nk, n3)), prep(v, (nk, n3)), prep(box, (nk,)), prep(dxdvdt, (nk, 3))
        self._settled(self.lib.nm_set_state(self.h, k0, nk, _dp(x), _dp(v), _dp(box), _dp(dxdvdt)))

    def get_state(self, k0=0, nk=None, velocities=True):
        """x, v, box, (dx, dv, dt) of slots [k0, k0+nk); velocities=False skips the copy of v (returned as None)"""
        nk = self.nslots - k0 if nk is None else nk
        n3 = 3 * self.natoms
        x = np.empty((nk, n3))
        v = np.empty((nk, n3)) if velocities else None
        box, d = np.empty(nk), np.empty((nk, 3))
        self._settled(self.lib.nm_get_state(self.h, k0, nk, _dp(x), _dp(v) if velocities else None, _dp(box), _dp(d)))
        return x, v, box, d

    def get_slots(self, slots, velocities=True):
        """x, v, box, (dx, dv, dt), thermo[5] of the listed local slots in one batch (nm_get_slots): one settle, one wait"""
        sl = np.ascontiguousarray(slots, dtype=np.int32)
        nk, n3 = len(sl), 3 * self.natoms
        x = np.empty((nk, n3))
        v = np.empty((nk, n3)) if velocities else None
        box, d, th = np.empty(nk), np.empty((nk, 3)), np.empty((nk, 5))
        self._settled(self.lib.nm_get_slots(self.h, nk, sl.ctypes.data_as(B.c_int_p), _dp(x), _dp(v) if velocities else None, _dp(box),
                                            _dp(d), _dp(th)))
        return x, v, box, d, th

    def set_slots(self, slots, x=None, v=None, box=None, dxdvdt=None, th=None):
        """the inverse of get_slots (nm_set_slots)"""
        sl = np.ascontiguousarray(slots, dtype=np.int32)
        nk, n3 = len(sl), 3 * self.natoms

        def prep(a, size):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=np.float64)
            if a.size != size:
                raise ValueError('bad array size %d, expected %d' % (a.size, size))
            return a
        x, v, box, dxdvdt, th = prep(x, nk * n3), prep(v, nk * n3), prep(box, nk), prep(dxdvdt, 3 * nk), prep(th, 5 * nk)
        self._settled(self.lib.nm_set_slots(self.h, nk, sl.ctypes.data_as(B.c_int_p), _dp(x), _dp(v), _dp(box), _dp(dxdvdt), _dp(th)))

    def init_lattice(self, dx=0.03125, dv=0.03125, interpolate=False):
        """init_samples (remcmc:394-456) through the C-ABI: the states lattice.init_states builds, without the Python front end"""
        self._chk(self.lib.nm_init_lattice(self.h, float(dx), float(dv), int(bool(interpolate))))

    def set_thermo(self, th, k0=0, nk=None):
        """th[nk][5] = temp, pe, ke, virial, vol (state-list entries 3,4,5,6,8): for restarts"""
        nk = self.nslots - k0 if nk is None else nk
        th = np.ascontiguousarray(th, dtype=np.float64)
        if th.size != 5 * nk:
            raise ValueError('bad thermo array')
        self._settled(self.lib.nm_set_thermo(self.h, k0, nk, _dp(th)))

    def set_step(self, step):
        self._chk(self.lib.nm_set_step(self.h, int(step)))

    # -- the hot path
    def run_block(self, mod):
        """gen_samples (remcmc:694-719): MOD moves for every replica, asynchronous"""
        self._chk(self.lib.nm_run_block(self.h, int(mod)))

    def run_cycles(self, ncycles, mod):
        """ncycles x (gen_samples, gen_mc_params, replica_exchange) with outputs off, steps STEP .. STEP + ncycles - 1 (nm_run_cycles): one launch in
        which only the replicas of a pressure row wait for one another, where the configuration has the kernel for it; the same chains either way"""
        self._chk(self.lib.nm_run_cycles(self.h, int(ncycles), int(mod)))

    def run_md(self, nsteps):
        """init_sample's -is dynamics (remcmc:421-425): velocities at T, then nsteps of NVE"""
        self._chk(self.lib.nm_run_md(self.h, int(nsteps)))

    def thermo(self):
        """rows[nslots][17] in the .thrm column order (remcmc:208)"""
        rows = np.empty((self.nslots, B.NM_THERMO_COLS))
        self._settled(self.lib.nm_get_thermo(self.h, _dp(rows)))
        return rows

    def snapshot(self):
        """keep what a recorded cycle writes as of this point of the queue (call right behind run_block, in front of adapt); the copy to the host
        runs beside the stream (nm_snapshot)"""
        self._chk(self.lib.nm_snapshot(self.h))

    def snapshot_fetch(self, positions=True):
        """(rows[nslots][17], x[nslots][3N] or None, box[nslots]) of the oldest pending snapshot; waits for its copy only (nm_snapshot_fetch)"""
        rows = np.empty((self.nslots, B.NM_THERMO_COLS))
        x = np.empty((self.nslots, 3 * self.natoms)) if positions else None
        box = np.empty(self.nslots)
        self._chk(self.lib.nm_snapshot_fetch(self.h, _dp(rows), _dp(x) if positions else None, _dp(box)))
        return rows, x, box

    def adapt(self):
        """gen_mc_params (remcmc:748-770)"""
        self._chk(self.lib.nm_adapt(self.h))

    def exchange(self, count=True):
        """replica_exchange (remcmc:776-803); returns the number of swaps when count=True"""
        if not count:
            self._chk(self.lib.nm_exchange(self.h, None))
            return None
        n = C.c_int(0)
        self._chk(self.lib.nm_exchange(self.h, C.byref(n)))
        return n.value

    def synchronize(self):
        self._settled(self.lib.nm_synchronize(self.h))  # (a re-issued block may have lowered cus_per_replica)

    def note(self):
        """what the residency probe gave up at creation and every block that had to be re-issued with fewer workgroups per
        replica since (nm_create_note); empty when nothing happened"""
        return self.lib.nm_create_note(self.h).decode()

    def status(self):
        """per-slot status bits (include/nm.h NM_ST_*) of the last block; 0 = fine"""
        st = np.empty(self.nslots, dtype=np.int32)
        self._settled(self.lib.nm_get_status(self.h, st.ctypes.data_as(B.c_int_p)))
        return st

    # -- measurement
    def timing_reset(self):
        self._chk(self.lib.nm_timing_reset(self.h))

    def timing(self):
        """(launches, total_ms) of the run_block kernel, from HIP events on the engine's stream"""
        n, ms = C.c_int(0), C.c_double(0.0)
        self._chk(self.lib.nm_timing_get(self.h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def stats(self, reset=False):
        """per-slot (evaluations, list rebuilds, energy evaluations, interacting pairs summed over those, block time in 100 MHz ticks,
        blocks handed over inside one XCD, blocks, HMC moves, longest list row built, list slots per atom) since the last reset"""
        s = np.empty((self.nslots, B.NM_STATS_COLS))
        self._settled(self.lib.nm_stats_get(self.h, _dp(s), int(reset)))
        return s

    # -- test-only
    def eval(self, forces=True):
        U, W = np.empty(self.nslots), np.empty(self.nslots)
        f = np.empty((self.nslots, 3 * self.natoms)) if forces else None
        self._chk(self.lib.nm_eval(self.h, _dp(U), _dp(W), _dp(f)))
        return U, W, f

    def set_rng_tape(self, tapes):
        """tapes: list of 1-D arrays (one per slot) or None"""
        if tapes is None:
            self._chk(self.lib.nm_set_rng_tape(self.h, None, None))
            return
        off = np.zeros(self.nslots + 1, dtype=np.int32)
        off[1:] = np.cumsum([len(t) for t in tapes])
        flat = np.ascontiguousarray(np.concatenate([np.asarray(t, dtype=np.float64) for t in tapes]))
        if flat.size == 0:
            flat = np.zeros(1)
        self._chk(self.lib.nm_set_rng_tape(self.h, _dp(flat), off.ctypes.data_as(B.c_int_p)))

    def set_exchange_tape(self, tape):
        if tape is None:
            self._chk(self.lib.nm_set_exchange_tape(self.h, None, 0))
            return
        t = np.ascontiguousarray(tape, dtype=np.float64)
        self._chk(self.lib.nm_set_exchange_tape(self.h, _dp(t), len(t)))

    def set_trace(self, on):
        self._chk(self.lib.nm_set_trace(self.h, int(bool(on))))

    def trace(self, mod):
        tr = np.empty((self.nslots, mod, B.NM_TRACE_COLS))
        self._settled(self.lib.nm_get_trace(self.h, _dp(tr), int(mod)))
        return tr

    def set_counters(self, count=None, ratio=None):
        """counters [nslots][6] and float32 ratios [nslots][3] as a block would have left them (input of adapt)"""
        cn = None if count is None else np.ascontiguousarray(count, dtype=np.float64)
        ra = None if ratio is None else np.ascontiguousarray(ratio, dtype=np.float32)
        self._settled(self.lib.nm_set_counters(self.h, _dp(cn), None if ra is None else ra.ctypes.data_as(B.c_float_p)))

    def perm(self):
        p = np.empty(self.nslots, dtype=np.int32)
        self._settled(self.lib.nm_get_perm(self.h, p.ctypes.data_as(B.c_int_p)))
        return p

    def exchange_crit(self):
        n = self.nrows * self.nt * (self.nt - 1) // 2
        c = np.empty(max(n, 1))
        self._settled(self.lib.nm_get_exchange_crit(self.h, _dp(c), n))
        return c[:n]
