import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
import neuralmelting_amd as nm
from neuralmelting_amd import lattice
P = np.linspace(1, 8, 8, dtype=np.float32); T = np.linspace(.25, 2.5, 8, dtype=np.float32)
x, v, box, d = lattice.init_states(4, P, T, 0.03125, 0.03125)
e = nm.Engine(256, P, T); e.set_state(x, v, box, d)
for s in range(30):
    e.set_step(s); e.run_block(128); e.adapt(); e.exchange(count=False)
e.synchronize(); e.stats(reset=True)
for s in range(30, 40):
    e.set_step(s); e.run_block(128); e.adapt(); e.exchange(count=False)
st = e.stats(); th = e.thermo()
blk = st[:, 4] / st[:, 6] * 1e-5
print('slot  T     rho    ms/block  maxrow  rebuilds/mv pairs/eval')
for k in range(64):
    print('%3d  %.3f  %.3f  %.3f   %3.0f   %.2f  %.0f' % (k, T[k % 8], 256 / th[k, 4], blk[k], st[k, 8], st[k, 1] / 1280, st[k, 3] / max(st[k, 2], 1)))
e.close()
