"""ctypes binding of the CPU ORACLE (oracle/nm_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under neuralmelting_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libnm_oracle.so')

c_double_p = C.POINTER(C.c_double)
c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)
c_u32_p = C.POINTER(C.c_uint32)


class BlockParams(C.Structure):
    _fields_ = [('mod', C.c_int), ('nstps', C.c_int), ('bulk', C.c_int), ('iter_revert', C.c_int),
                ('ppos', C.c_double), ('pvol', C.c_double), ('lat', C.c_double), ('t', C.c_double),
                ('et', C.c_double), ('pf', C.c_double), ('tape', c_double_p), ('tape_len', C.c_int),
                ('trace', c_double_p)]


def build(force=False):
    """compile the oracle with gcc (building the checker is not using it)"""
    src = os.path.join(_HERE, 'nm_oracle.c')
    if force or not os.path.isfile(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s'])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(_SO):
        build()
    L = C.CDLL(_SO)
    vp = C.c_void_p
    L.orc_create.restype = vp
    L.orc_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int]
    L.orc_destroy.argtypes = [vp]
    L.orc_set_rng.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32]
    L.orc_set_box.argtypes = [vp, C.c_double]
    L.orc_get_box.restype = C.c_double
    L.orc_get_box.argtypes = [vp]
    for name in ('orc_set_x', 'orc_set_v', 'orc_get_x', 'orc_get_v', 'orc_get_f'):
        getattr(L, name).argtypes = [vp, c_double_p]
    L.orc_get_image.argtypes = [vp, c_int_p]
    L.orc_set_image.argtypes = [vp, c_int_p]
    L.orc_setup.argtypes = [vp]
    L.orc_eval_allpairs.argtypes = [vp, c_double_p, c_double_p, c_double_p]
    L.orc_displace.argtypes = [vp, C.c_double, C.c_uint32]
    L.orc_velocity_create.argtypes = [vp, C.c_double, C.c_uint32]
    L.orc_zero_linear.argtypes = [vp]
    L.orc_zero_angular.argtypes = [vp]
    L.orc_set_timestep.argtypes = [vp, C.c_double]
    L.orc_run.argtypes = [vp, C.c_int]
    for name in ('orc_pe', 'orc_ke', 'orc_temp', 'orc_press', 'orc_virial'):
        getattr(L, name).restype = C.c_double
        getattr(L, name).argtypes = [vp]
    L.orc_nlist.argtypes = [vp]
    L.orc_npairs.argtypes = [vp]
    L.orc_q6.restype = C.c_double
    L.orc_q6.argtypes = [C.c_double]
    L.orc_q6_printf.restype = C.c_double
    L.orc_q6_printf.argtypes = [C.c_double]
    L.orc_philox4x32_10.argtypes = [c_u32_p, c_u32_p, c_u32_p]
    L.orc_u01.restype = C.c_double
    L.orc_u01.argtypes = [C.c_uint32, C.c_uint32]
    L.orc_run_block.argtypes = [vp, C.POINTER(BlockParams), c_double_p, c_double_p, c_double_p, c_double_p,
                                c_double_p, c_double_p, c_float_p, c_int_p]
    L.orc_run_blocks.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32,
                                 C.POINTER(BlockParams), c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                 c_double_p, c_float_p, C.c_int]
    L.orc_adapt.argtypes = [c_float_p, c_double_p]
    L.orc_exchange.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, c_double_p, c_double_p,
                               c_double_p, c_double_p, c_int_p, c_double_p, c_double_p]
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def q6(x):
    return lib().orc_q6(float(x))


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return [int(v) for v in o]


class Sim:
    """one LAMMPS-like instance of the oracle (the primitives the reference's call sites use)"""

    def __init__(self, natoms, units=0, mass=1.0, pot=0):
        self.L = lib()
        self.n = natoms
        self.h = self.L.orc_create(natoms, units, mass, pot)

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_rng(self, seed, slot, step):
        self.L.orc_set_rng(self.h, seed, slot, step)

    def set_box(self, box):
        self.L.orc_set_box(self.h, float(box))

    def get_box(self):
        return self.L.orc_get_box(self.h)

    def set_x(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        self.L.orc_set_x(self.h, _dp(x))

    def set_v(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        self.L.orc_set_v(self.h, _dp(v))

    def get_x(self):
        x = np.empty(3 * self.n)
        self.L.orc_get_x(self.h, _dp(x))
        return x

    def get_v(self):
        v = np.empty(3 * self.n)
        self.L.orc_get_v(self.h, _dp(v))
        return v

    def get_f(self):
        f = np.empty(3 * self.n)
        self.L.orc_get_f(self.h, _dp(f))
        return f

    def setup(self):
        rc = self.L.orc_setup(self.h)
        if rc:
            raise RuntimeError('oracle setup failed: %d' % rc)

    def eval_allpairs(self, forces=True):
        U, W = C.c_double(), C.c_double()
        f = np.zeros(3 * self.n)
        rc = self.L.orc_eval_allpairs(self.h, C.byref(U), C.byref(W), _dp(f) if forces else None)
        if rc:
            raise RuntimeError('oracle allpairs failed')
        return U.value, W.value, f

    def displace(self, a, tag):
        self.L.orc_displace(self.h, float(a), int(tag))

    def velocity_create(self, t, tag):
        self.L.orc_velocity_create(self.h, float(t), int(tag))

    def zero_linear(self):
        self.L.orc_zero_linear(self.h)

    def zero_angular(self):
        self.L.orc_zero_angular(self.h)

    def set_timestep(self, h):
        self.L.orc_set_timestep(self.h, float(h))

    def run(self, n):
        rc = self.L.orc_run(self.h, int(n))
        if rc:
            raise RuntimeError('oracle run failed: %d' % rc)

    pe = property(lambda self: self.L.orc_pe(self.h))
    ke = property(lambda self: self.L.orc_ke(self.h))
    temp = property(lambda self: self.L.orc_temp(self.h))
    press = property(lambda self: self.L.orc_press(self.h))
    virial = property(lambda self: self.L.orc_virial(self.h))
    nlist = property(lambda self: self.L.orc_nlist(self.h))
    npairs = property(lambda self: self.L.orc_npairs(self.h))

    def run_block(self, x, v, box, dxdvdt, *, mod, nstps, bulk, ppos, pvol, lat, t, et, pf,
                  iter_revert=0, tape=None, trace=False):
        """gen_sample (remcmc:665-691) for this replica; returns dict of outputs"""
        n = self.n
        x = np.array(x, dtype=np.float64).reshape(3 * n).copy()
        v = np.array(v, dtype=np.float64).reshape(3 * n).copy()
        boxc = C.c_double(float(box))
        dxdvdt = np.array(dxdvdt, dtype=np.float64)
        thermo = np.zeros(5)
        counters = np.zeros(6)
        ratios = np.zeros(3, dtype=np.float32)
        p = BlockParams()
        p.mod, p.nstps, p.bulk, p.iter_revert = mod, nstps, int(bulk), int(iter_revert)
        p.ppos, p.pvol, p.lat, p.t, p.et, p.pf = ppos, pvol, lat, t, et, pf
        tp = None
        if tape is not None:
            tp = np.ascontiguousarray(tape, dtype=np.float64)
            p.tape, p.tape_len = _dp(tp), len(tp)
        tr = None
        if trace:
            tr = np.zeros((mod, 4))
            p.trace = _dp(tr)
        used = C.c_int(0)
        rc = self.L.orc_run_block(self.h, C.byref(p), _dp(x), _dp(v), C.byref(boxc), _dp(dxdvdt), _dp(thermo),
                                  _dp(counters), ratios.ctypes.data_as(c_float_p), C.byref(used))
        if rc:
            raise RuntimeError('oracle run_block failed: %d' % rc)
        return dict(x=x, v=v, box=boxc.value, thermo=thermo, counters=counters, ratios=ratios, trace=tr,
                    tape_used=used.value)


def run_blocks(x, v, box, dxdvdt, t, et, pf, *, natoms, mod, nstps, bulk, ppos, pvol, lat, seed, slot0, step,
               units=0, mass=1.0, pot=0, iter_revert=0, nthreads=0):
    """gen_samples (remcmc:694-719) over ns replicas, fanned out with OpenMP"""
    ns = len(box)
    x = np.array(x, dtype=np.float64).reshape(ns, 3 * natoms).copy()
    v = np.array(v, dtype=np.float64).reshape(ns, 3 * natoms).copy()
    box = np.array(box, dtype=np.float64).copy()
    dxdvdt = np.ascontiguousarray(dxdvdt, dtype=np.float64).reshape(ns, 3)
    thermo = np.zeros((ns, 5))
    counters = np.zeros((ns, 6))
    ratios = np.zeros((ns, 3), dtype=np.float32)
    ps = (BlockParams * ns)()
    for k in range(ns):
        ps[k].mod, ps[k].nstps, ps[k].bulk, ps[k].iter_revert = mod, nstps, int(bulk), int(iter_revert)
        ps[k].ppos, ps[k].pvol, ps[k].lat = ppos, pvol, lat
        ps[k].t, ps[k].et, ps[k].pf = float(t[k]), float(et[k]), float(pf[k])
    rc = lib().orc_run_blocks(ns, natoms, units, mass, pot, seed, slot0, step, ps, _dp(x), _dp(v), _dp(box),
                              _dp(dxdvdt), _dp(thermo), _dp(counters), ratios.ctypes.data_as(c_float_p), nthreads)
    if rc:
        raise RuntimeError('oracle run_blocks failed: %d' % rc)
    return dict(x=x, v=v, box=box, thermo=thermo, counters=counters, ratios=ratios)


def adapt(ratios, dxdvdt):
    r = np.array(ratios, dtype=np.float32)
    d = np.array(dxdvdt, dtype=np.float64)
    lib().orc_adapt(r.ctypes.data_as(c_float_p), _dp(d))
    return d


def exchange(np_total, nt, row0, nrows, seed, step, etot, vol, et, pf, perm=None, tape=None):
    """replica_exchange (remcmc:776-803) on local rows; returns (swaps, perm, etot, vol, crit)"""
    ns = nrows * nt
    etot = np.array(etot, dtype=np.float64).copy()
    vol = np.array(vol, dtype=np.float64).copy()
    et = np.ascontiguousarray(et, dtype=np.float64)
    pf = np.ascontiguousarray(pf, dtype=np.float64)
    perm = np.arange(ns, dtype=np.int32) if perm is None else np.array(perm, dtype=np.int32).copy()
    npairs = nrows * nt * (nt - 1) // 2
    crit = np.zeros(max(npairs, 1))
    tp = None
    if tape is not None:
        tp = np.ascontiguousarray(tape, dtype=np.float64)
    swaps = lib().orc_exchange(np_total, nt, row0, nrows, seed, step, _dp(etot), _dp(vol), _dp(et), _dp(pf),
                               perm.ctypes.data_as(c_int_p), _dp(tp) if tp is not None else None, _dp(crit))
    return swaps, perm, etot, vol, crit[:npairs]
