"""dev probe: C2 throughput with the reference's default position move (iterative, one trial per atom) against bulk"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import neuralmelting_amd as nm
from neuralmelting_amd import lattice

P = np.linspace(1, 8, 8, dtype=np.float32); T = np.linspace(.25, 2.5, 8, dtype=np.float32)
x, v, box, d = lattice.init_states(4, P, T, 0.03125, 0.03125)
for bulk in (True, False):
    e = nm.Engine(256, P, T, bulk=bulk)
    e.set_state(x, v, box, d)
    for s in range(4):
        e.set_step(s); e.run_block(128); e.adapt(); e.exchange(count=False)
    e.synchronize(); e.timing_reset()
    t0 = time.perf_counter()
    for s in range(4, 10):
        e.set_step(s); e.run_block(128); e.adapt(); e.exchange(count=False)
    e.synchronize()
    dt = time.perf_counter() - t0
    n, ms = e.timing()
    print('bulk' if bulk else 'iterative', '%.0f sweeps/s, kernel %.2f ms per launch' % (64 * 128 * 6 / dt, ms / n))
    e.close()
