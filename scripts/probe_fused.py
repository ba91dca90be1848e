"""nm_run_cycles against the loop of nm_run_block + nm_adapt + nm_exchange on the 8 x 8 C2 grid: wall time of each and whether thermo rows, positions,
velocities and the slot permutation are equal bit for bit (the quick check used while nm_cycles_kernel was built; the tests proper are
tests/test_cycles_gpu.py).  NM_FUSED_CYCLES=0 / all select the launch shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import neuralmelting_amd as nm
from neuralmelting_amd import lattice
P = np.linspace(1, 8, 8, dtype=np.float32); T = np.linspace(.25, 2.5, 8, dtype=np.float32)
x, v, box, d = lattice.init_states(4, P, T, 0.03125, 0.03125)
def run(fused, K=6, mod=16):
    e = nm.Engine(256, P, T); e.set_state(x, v, box, d)
    e.set_step(0)
    t0 = time.perf_counter()
    if fused:
        e.run_cycles(K, mod)
    else:
        for s in range(K):
            e.set_step(s); e.run_block(mod); e.adapt(); e.exchange(count=False)
    e.synchronize()
    dt = time.perf_counter() - t0
    out = (e.thermo(), e.get_state(), e.perm(), e.note())
    e.close()
    return out, dt
(a, ta), (b, tb) = run(False), run(True)
print('single %.3f s fused %.3f s' % (ta, tb), 'note', repr(b[3]))
print('thermo equal', np.array_equal(a[0], b[0]), 'x equal', np.array_equal(a[1][0], b[1][0]), 'v equal', np.array_equal(a[1][1], b[1][1]), 'perm equal', np.array_equal(a[2], b[2]))
if not np.array_equal(a[0], b[0]):
    bad = np.argwhere(a[0] != b[0]); print(bad[:10]); print(a[0][bad[0][0]], b[0][bad[0][0]])
