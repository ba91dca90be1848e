"""Does the order of the atoms in memory matter to the pair loops?  The same replicas, once with the atoms in lattice order
(neighbouring indices = neighbouring sites, what lattice.init_states makes) and once with the indices shuffled: same physics, same
list lengths, different LDS addresses per wave instruction.  Cold grid: T* <= 0.6, the crystals stay crystals and keep their
order; hot grid: liquids, whose atoms leave their sites anyway.

    python scripts/probe_locality.py [config cycles]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import neuralmelting_amd as nm
from neuralmelting_amd import lattice
import bench


def run(config, cycles, shuffle, hot):
    el, sz, rows, np_cfg, tn, mod, _ = bench.CONFIGS[config]
    P = np.linspace(1.0, 8.0, rows, dtype=np.float32)
    if el == 'LJ':
        T = np.linspace(1.8, 2.5, tn, dtype=np.float32) if hot else np.linspace(0.25, 0.6, tn, dtype=np.float32)
    else:
        T = np.linspace(1800.0, 2560.0, tn, dtype=np.float32) if hot else np.linspace(256.0, 600.0, tn, dtype=np.float32)
    x, v, box, d = lattice.init_states(sz, P, T, 0.03125, 0.03125, el=el, row0=0, nrows=rows)
    if shuffle:
        n = x.shape[1] // 3   # (slots, 3 N), atom by atom (gather_atoms order)
        perm = np.random.default_rng(7).permutation(n)
        x = np.ascontiguousarray(x.reshape(-1, n, 3)[:, perm]).reshape(-1, 3 * n); v = np.ascontiguousarray(v.reshape(-1, n, 3)[:, perm]).reshape(-1, 3 * n)
    e = nm.Engine(4 * sz ** 3, P, T, element=el, row0=0, nrows=rows)
    e.set_state(x, v, box, d)
    warm = 30 if hot else 4
    for s in range(warm):
        e.set_step(s); e.run_block(mod); e.adapt(); e.exchange(count=False)
    e.synchronize(); e.timing_reset(); e.stats(reset=True)
    for s in range(warm, warm + cycles):
        e.set_step(s); e.run_block(mod); e.adapt(); e.exchange(count=False)
    n, ms = e.timing(); st = e.stats()
    print('%s %-5s %-9s Q=%d  %.3f ms per launch; evals/move %.2f rebuilds/move %.2f pairs/eval %.0f'
          % (config, 'hot' if hot else 'cold', 'shuffled' if shuffle else 'ordered', e.cus_per_replica, ms / n,
             st[:, 0].sum() / (mod * cycles * e.nslots), st[:, 1].sum() / (mod * cycles * e.nslots), st[:, 3].sum() / max(st[:, 2].sum(), 1)), flush=True)
    e.close()


if __name__ == '__main__':
    a = sys.argv[1:]
    cfg = a[0] if a else 'C2'; cyc = int(a[1]) if len(a) > 1 else 10
    for hot in (False, True):
        for sh in (False, True):
            run(cfg, cyc, sh, hot)
