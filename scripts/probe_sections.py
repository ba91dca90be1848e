"""dev probe (diagnostic build libnm_hip_prof.so): shader-clock share of each section of the block kernel"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['NM_HIP_LIB'] = os.path.join(ROOT, 'neuralmelting_amd', os.environ.get('NM_PROF_LIB', 'libnm_hip_prof.so'))
import numpy as np
import neuralmelting_amd as nm
from neuralmelting_amd import lattice, _lib

NAMES = ['eval:entry barrier', 'eval:list check', 'eval:rebuild', 'eval:pair loop', 'eval:reduce/barrier',
         'post:init', 'post:bulk', 'post:vmc', 'post:hmc start', 'post:hmc step + hand-over', 'pre:bulk', 'pre:vmc', 'pre:hmc', 'iter pmc', 'eval:cluster exchange', '-']

def main(sz=4, rows=8, tn=8, mod=128, cycles=3, warm=6):
    npn = int(os.environ.get('NP_ALL', rows))
    el = os.environ.get('NM_PROBE_EL', 'LJ')   # NM_PROBE_EL=Al: the EAM kernels (BASELINE config 4)
    P = np.linspace(1, 8, npn, dtype=np.float32)
    T = np.linspace(.25, 2.5, tn, dtype=np.float32) if el == 'LJ' else np.linspace(256.0, 2560.0, tn, dtype=np.float32)
    x, v, box, d = lattice.init_states(sz, P, T, 0.03125, 0.03125, el=el, row0=0, nrows=rows)
    e = nm.Engine(4 * sz ** 3, P, T, element=el, row0=0, nrows=rows, bulk=os.environ.get('NM_PROBE_ITER') != '1')  # NM_PROBE_ITER=1: iterative position moves
    e.set_state(x, v, box, d)
    L = _lib.load()
    L.nm_prof_get.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    buf = np.zeros((e.nslots, 16), dtype=np.uint64)
    for s in range(warm):
        e.set_step(s); e.run_block(mod); e.adapt(); e.exchange(count=False)
    L.nm_prof_get(e.h, buf.ctypes.data, 1)
    e.timing_reset(); e.stats(reset=True)
    for s in range(warm, warm + cycles):
        e.set_step(s); e.run_block(mod); e.adapt(); e.exchange(count=False)
    L.nm_prof_get(e.h, buf.ctypes.data, 1)
    n, ms = e.timing(); st = e.stats()
    moves = mod * cycles
    tot = buf.sum(1).astype(float)
    k = int(np.argmax(tot))
    print('kernel %.2f ms/launch; slowest slot %d: %.0f cycles/move stamped (%.1f us at 2.4 GHz); evals/move %.2f rebuilds/move %.2f'
          % (ms / n, k, tot[k] / moves, tot[k] / moves / 2400.0, st[k, 0] / moves, st[k, 1] / moves))
    mean = buf.mean(0) / moves
    for q, name in enumerate(NAMES):
        if mean[q] > 0:
            print('  %-22s mean %8.0f cyc/move (%5.1f%%)   slowest slot %8.0f' % (name, mean[q], 100 * mean[q] / mean.sum(), buf[k, q] / moves))
    ev = st[:, 0].mean() / moves
    print('  per evaluation: pair loop %.0f cyc, entry %.0f, check %.0f, reduce %.0f; per rebuild %.0f cyc'
          % (mean[3] / ev, mean[0] / ev, mean[1] / ev, mean[4] / ev, buf[:, 2].sum() / max(st[:, 1].sum(), 1)))
    e.close()

if __name__ == '__main__':
    # python scripts/probe_sections.py [sz rows tn mod cycles warm]
    a = [int(v) for v in sys.argv[1:]]
    main(*a)
