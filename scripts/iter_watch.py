"""dev probe: the reference's iterative position move on the C2 grid, cycle by cycle: when do the chains leave the floating-point range?
    python scripts/iter_watch.py [cycles]        (run from the repo root, or from a worktree of an older commit)"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import neuralmelting_amd as nm
from neuralmelting_amd import lattice
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 40
P = np.linspace(1.0, 8.0, 8, dtype=np.float32); T = np.linspace(0.25, 2.5, 8, dtype=np.float32)
x, v, box, d = lattice.init_states(4, P, T, 0.03125, 0.03125)
e = nm.Engine(256, P, T, bulk=False)
e.set_state(x, v, box, d)
for step in range(cycles):
    e.set_step(step); e.run_block(128)
    try:
        rows = e.thermo()
    except nm.NMError as err:
        print('cycle', step, 'ERROR', str(err)[-120:]); break
    st = e.stats(reset=True)
    k = int(np.argmax(rows[:, 1]))
    print('cycle %2d  max pe/N %10.4g (slot %2d)  dx max %.3f  acc_pmc mean %.2f  acc_hmc mean %.2f  rebuilds/sweep %.2f  slot35: pe/N %.4g dx %.3f ap %.2f'
          % (step, rows[k, 1] / 256, k, rows[:, 5].max(), rows[:, 14].mean(), rows[:, 16].mean(), st[:, 1].sum() / (64 * 128),
             rows[35, 1] / 256, rows[35, 5], rows[35, 14]), flush=True)
    e.adapt(); e.exchange(count=False)
e.close()
