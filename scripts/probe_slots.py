"""dev probe: per-cycle launch time and per-slot work counters of a bench preset (which replicas set the pace of a launch)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import neuralmelting_amd as nm
from neuralmelting_amd import lattice
import bench

def main(cfg='C5', cycles=18):
    el, sz, rows, npn, tn, mod, desc = bench.CONFIGS[cfg]
    P = np.linspace(1.0, 8.0, npn, dtype=np.float32)
    T = np.linspace(0.25, 2.5, tn, dtype=np.float32) if el == 'LJ' else np.linspace(256.0, 2560.0, tn, dtype=np.float32)
    x, v, box, d = lattice.init_states(sz, P, T, 0.03125, 0.03125, el=el, row0=0, nrows=rows)
    e = nm.Engine(4 * sz ** 3, P, T, element=el, row0=0, nrows=rows)
    e.set_state(x, v, box, d)
    for s in range(cycles):
        e.timing_reset(); e.stats(reset=True)
        e.set_step(s); e.run_block(mod); e.synchronize()
        n, ms = e.timing(); st = e.stats(); r = e.thermo()
        e.adapt(); e.exchange(count=False)
        k = int(np.argmax(st[:, 1]))
        print('cycle %2d: %.1f ms; rebuilds/move mean %.3f max %.3f (slot %d, T index %d); evals/move %.2f; hmc acc %.2f; dt mean %.5f max %.5f; pairs %.0f'
              % (s, ms, st[:, 1].mean() / mod, st[:, 1].max() / mod, k, k % tn, st[:, 0].mean() / mod, r[:, 16].mean(), r[:, 7].mean(), r[:, 7].max(),
                 st[:, 3].sum() / max(st[:, 2].sum(), 1)))
    e.close()

if __name__ == '__main__':
    main(*(sys.argv[1:2] or ['C5']), **({'cycles': int(sys.argv[2])} if len(sys.argv) > 2 else {}))
