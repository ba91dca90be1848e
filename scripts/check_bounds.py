"""Diagnostic build (make prof) only: every index into the global spill / list arrays of the large-cell kernels is checked inside
the kernel (nm_kernels.h NM_CHECK_INDEX: counted and redirected, never dereferenced out of range).  This runs the kernel
instantiations that use those arrays — 5^3 / 6^3 at 1, 2, 4, 8 workgroups per replica, 8^3 at 1 and 2, bulk and iterative position
moves, HMC-heavy blocks whose rejected trajectories go back to the saved copies and the second list — and prints the count.

    NM_HIP_LIB=$PWD/neuralmelting_amd/libnm_hip_prof.so python scripts/check_bounds.py
"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

os.environ['NM_DBG'] = '8'   # switches the exact list self-check of the diagnostic build on


def main():
    import neuralmelting_amd as nm
    from neuralmelting_amd import lattice, _lib
    assert 'prof' in _lib.LIB_PATH, 'run with NM_HIP_LIB pointing at libnm_hip_prof.so'
    L = _lib.load()
    total = 0
    P = np.linspace(1.0, 8.0, 2, dtype=np.float32)
    T = np.linspace(0.25, 2.5, 4, dtype=np.float32)
    for sz, qs in ((5, (1, 2, 4, 8)), (6, (1, 2, 4, 8)), (8, (1, 2, 4))):
        for q in qs:
            for bulk in (True, False):
                os.environ['NM_CUS_PER_REPLICA'] = str(q)
                x, v, box, d = lattice.init_states(sz, P, T, 0.03125, 0.03125)
                d[:, 2] = 0.002                      # short steps: trajectories get accepted as well as rejected
                e = nm.Engine(4 * sz ** 3, P, T, bulk=bulk, ppos=0.2, pvol=0.2)
                assert e.cus_per_replica == q
                e.set_state(x, v, box, d)
                for step in range(3):
                    e.set_step(step); e.run_block(12); e.adapt(); e.exchange(count=False)
                e.synchronize()
                st = e.stats()
                n, m = C.c_uint(0), C.c_uint(0)
                assert L.nm_prof_oob(e.h, C.byref(n)) == 0 and L.nm_prof_list_miss(e.h, C.byref(m)) == 0
                print('%d^3 Q=%d %-9s rebuilds %4d  evaluations %5d  out-of-range indices %d  incomplete list rows %d'
                      % (sz, q, 'bulk' if bulk else 'iterative', st[:, 1].sum(), st[:, 0].sum(), n.value, m.value), flush=True)
                total += n.value + m.value
                e.close()
    # the 4^3 byte lists (LDS, kept twice), every workgroups-per-replica setting, LJ and the EAM, equilibrated chains incl. the
    # reference's never-undone iterative trials: every rebuild checked against exact separations
    T8 = np.linspace(0.25, 2.5, 8, dtype=np.float32)
    for el, qs in (('LJ', (1, 2, 4, 8)), ('Al', (1, 2, 4))):
        for q in qs:
            for bulk in (True, False):
                os.environ['NM_CUS_PER_REPLICA'] = str(q)
                Tg = T8 if el == 'LJ' else np.linspace(256.0, 2560.0, 8, dtype=np.float32)
                Pg = np.linspace(1.0, 8.0, 2, dtype=np.float32)
                x, v, box, d = lattice.init_states(4, Pg, Tg, 0.03125, 0.03125, el=el)
                e = nm.Engine(256, Pg, Tg, element=el, bulk=bulk)
                assert e.cus_per_replica == q
                e.set_state(x, v, box, d)
                cycles = 40 if bulk else 10
                try:
                    for step in range(cycles):
                        e.set_step(step); e.run_block(64); e.adapt(); e.exchange(count=False)
                    e.synchronize()
                    note = ''
                except nm.NMError as err:
                    note = ' [' + str(err)[-60:] + ']'
                st = e.stats()
                n, m = C.c_uint(0), C.c_uint(0)
                assert L.nm_prof_oob(e.h, C.byref(n)) == 0 and L.nm_prof_list_miss(e.h, C.byref(m)) == 0
                print('4^3 %s Q=%d %-9s rebuilds %5d  evaluations %6d  out-of-range indices %d  incomplete list rows %d%s'
                      % (el, q, 'bulk' if bulk else 'iterative', st[:, 1].sum(), st[:, 0].sum(), n.value, m.value, note), flush=True)
                if m.value:
                    info = np.zeros(16)
                    L.nm_prof_miss_info(e.h, info.ctypes.data_as(C.POINTER(C.c_double)))
                    print('    first incomplete row: slot %d atom %d lacks atom %d; exact count %d, listed inside %d, row length %d, L %.6f, radius %.3f'
                          % tuple(info[:8]))
                    if info[2] >= 0:
                        xi, xj, Lb = info[8:11], info[11:14], info[6]
                        dd = xi - xj; dd -= Lb * np.rint(dd / Lb)
                        print('    x_i', xi, 'x_j', xj, 'separation %.9f' % np.sqrt((dd * dd).sum()), 'fixed i %x j %x' % (int(info[14]), int(info[15])))
                total += n.value + m.value
                e.close()
    print('TOTAL out-of-range indices + incomplete list rows: %d' % total)
    return 1 if total else 0


if __name__ == '__main__':
    sys.exit(main())
