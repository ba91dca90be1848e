"""dev probe (build `make ab8`): start and duration of every slot's block in one launch of a grid that runs in rounds (more clusters
than the chip holds): how well do the rounds pack?   python scripts/probe_rounds.py [config warm]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['NM_HIP_LIB'] = os.path.join(ROOT, 'neuralmelting_amd', 'libnm_hip_ab8.so')
import numpy as np
import neuralmelting_amd as nm
from neuralmelting_amd import lattice
import bench

config = sys.argv[1] if len(sys.argv) > 1 else 'C5'
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 30
el, sz, rows, np_cfg, tn, mod, _ = bench.CONFIGS[config]
P = np.linspace(1.0, 8.0, np_cfg, dtype=np.float32)
T = np.linspace(0.25, 2.5, tn, dtype=np.float32)
x, v, box, d = lattice.init_states(sz, P, T, 0.03125, 0.03125, el=el, row0=0, nrows=rows)
e = nm.Engine(4 * sz ** 3, P, T, element=el, row0=0, nrows=rows)
e.set_state(x, v, box, d)
for s in range(warm):
    e.set_step(s); e.run_block(mod); e.adapt(); e.exchange(count=False)
e.synchronize(); e.timing_reset(); e.stats(reset=True)
e.set_step(warm); e.run_block(mod); e.synchronize()
n, ms = e.timing(); st = e.stats()
start = (st[:, 9] - st[:, 9].min()) * 1e-5; dur = st[:, 4] * 1e-5
end = start + dur
print('%s: Q = %d, %d slots; kernel %.2f ms; block durations mean %.2f max %.2f; sum x Q / 256 CUs = %.2f ms; last end %.2f ms'
      % (config, e.cus_per_replica, e.nslots, ms / n, dur.mean(), dur.max(), dur.sum() * e.cus_per_replica / 256.0, end.max()))
o = np.argsort(start)
print('slot  start   dur    end   (ms, by start)')
for k in o:
    print('%4d %6.2f %6.2f %6.2f' % (k, start[k], dur[k], end[k]))
late = start > 1.0
print('second round: %d slots, starts %.2f .. %.2f ms' % (late.sum(), start[late].min() if late.any() else 0, start[late].max() if late.any() else 0))
e.close()
