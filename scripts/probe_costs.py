"""dev probe: split the block kernel's time into per-move and per-evaluation costs by running move mixes"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import neuralmelting_amd as nm
from neuralmelting_amd import lattice

def run(ppos, pvol, nstps, mod=128, cycles=4, sz=4, rows=8, tn=8, warm=4, bulk=True):
    P = np.linspace(1, 8, rows, dtype=np.float32); T = np.linspace(.25, 2.5, tn, dtype=np.float32)
    x, v, box, d = lattice.init_states(sz, P, T, 0.03125, 0.03125)
    e = nm.Engine(4 * sz ** 3, P, T, ppos=ppos, pvol=pvol, nstps=nstps, bulk=bulk)
    e.set_state(x, v, box, d)
    for s in range(warm):
        e.set_step(s); e.run_block(mod); e.adapt(); e.exchange(count=False)
    e.synchronize(); e.timing_reset(); e.stats(reset=True)
    for s in range(warm, warm + cycles):
        e.set_step(s); e.run_block(mod); e.adapt(); e.exchange(count=False)
    e.synchronize()
    n, ms = e.timing(); st = e.stats()
    moves = mod * cycles
    per_move_us = ms / moves * 1e3
    ev = st[:, 0].mean() / moves; rb = st[:, 1].mean() / moves
    print('ppos %.2f pvol %.2f nstps %2d bulk %d: %.1f us/move (slowest replica), evals/move %.2f (max %.2f) rebuilds/move %.3f (max %.3f) pairs %.0f'
          % (ppos, pvol, nstps, bulk, per_move_us, ev, st[:, 0].max() / moves, rb, st[:, 1].max() / moves,
             st[:, 3].sum() / max(st[:, 2].sum(), 1)))
    e.close()
    return per_move_us

if __name__ == '__main__':
    run(1.0, 0.0, 8)
    run(0.0, 1.0, 8)
    run(0.0, 0.0, 1)
    run(0.0, 0.0, 8)
    run(0.0, 0.0, 16)
    run(0.125, 0.125, 8)
