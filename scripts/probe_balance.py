"""How evenly do the replicas of a launch finish?  A launch lasts as long as its slowest replica; this prints, per slot, the
time its blocks took (nm_stats_get column 4: 100 MHz ticks from kernel entry to the exit of the replica's first workgroup) next to
the work they did, after `warm` cycles of equilibration.

    python scripts/probe_balance.py [config warm cycles [rows]]      e.g. C2 30 10, or C2 30 10 4 (the first 4 of the preset's rows)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import neuralmelting_amd as nm
from neuralmelting_amd import lattice
import bench


def main(config='C2', warm=30, cycles=10, rows_override=None):
    el, sz, rows, np_cfg, tn, mod, _ = bench.CONFIGS[config]
    if rows_override:   # a share of the preset's grid: its first `rows_override` pressure rows
        rows = rows_override
    P = np.linspace(1.0, 8.0, np_cfg if rows_override else rows, dtype=np.float32)
    T = np.linspace(0.25, 2.5, tn, dtype=np.float32) if el == 'LJ' else np.linspace(256.0, 2560.0, tn, dtype=np.float32)
    x, v, box, d = lattice.init_states(sz, P, T, 0.03125, 0.03125, el=el, row0=0, nrows=rows)
    e = nm.Engine(4 * sz ** 3, P, T, element=el, row0=0, nrows=rows)
    e.set_state(x, v, box, d)
    for s in range(warm):
        e.set_step(s); e.run_block(mod); e.adapt(); e.exchange(count=False)
    e.synchronize(); e.timing_reset(); e.stats(reset=True)
    acc = np.zeros((e.nslots, 3))
    for s in range(warm, warm + cycles):
        e.set_step(s); e.run_block(mod)
        acc += e.thermo()[:, 14:17]
        e.adapt(); e.exchange(count=False)
    n, ms = e.timing(); st = e.stats()
    blk_ms = st[:, 4] / st[:, 6] * 1e-5  # ticks of 10 ns -> ms
    moves = mod * cycles
    print('%s: Q = %d, kernel %.3f ms per launch; per-slot block time mean %.3f  max %.3f  min %.3f ms (max / mean %.3f); one-XCD clusters %.0f %%'
          % (config, e.cus_per_replica, ms / n, blk_ms.mean(), blk_ms.max(), blk_ms.min(), blk_ms.max() / blk_ms.mean(),
             100.0 * st[:, 5].sum() / st[:, 6].sum()))
    print('slot  P     T      ms/block  evals/mv  rebuilds/mv  pairs/eval  hmc/block  acc(p v h)')
    for k in range(e.nslots):
        i, j = divmod(k, tn)
        print('%4d  %.2f  %.3f  %.3f     %.2f      %.2f         %6.0f      %.1f      %.2f %.2f %.2f'
              % (k, P[i], T[j], blk_ms[k], st[k, 0] / moves, st[k, 1] / moves, st[k, 3] / max(st[k, 2], 1), st[k, 7] / st[k, 6],
                 acc[k, 0] / cycles, acc[k, 1] / cycles, acc[k, 2] / cycles))
    # what would a launch cost if every replica took the mean / if rows were balanced
    per_row = blk_ms.reshape(rows, tn)
    print('row maxima (ms):', ' '.join('%.3f' % m for m in per_row.max(1)))
    print('column (temperature) means (ms):', ' '.join('%.3f' % m for m in per_row.mean(0)))
    e.close()


if __name__ == '__main__':
    a = sys.argv[1:]
    main(a[0] if a else 'C2', int(a[1]) if len(a) > 1 else 30, int(a[2]) if len(a) > 2 else 10, int(a[3]) if len(a) > 3 else None)
