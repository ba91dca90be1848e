"""dev probe: what of a recorded cycle costs the block kernel its 4 %?  C2, equilibrated; per variant the mean HIP-event time of the block kernel and the
wall time per cycle: (a) outputs off, (b) snapshot + fetch one cycle later, nothing written, (c) the same + the files written by a helper thread"""
import os, sys, time, threading, tempfile, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if os.environ.get("NM_PROBE_TORCH"):
    import torch; torch.cuda.synchronize()
import neuralmelting_amd as nm
from neuralmelting_amd import lattice, _lib as B
P = np.linspace(1, 8, 8, dtype=np.float32); T = np.linspace(.25, 2.5, 8, dtype=np.float32)
x, v, box, d = lattice.init_states(4, P, T, 0.03125, 0.03125)
e = nm.Engine(256, P, T); e.set_state(x, v, box, d)
step = 0
for _ in range(30):
    e.set_step(step); e.run_block(128); e.adapt(); e.exchange(count=False); step += 1
td = tempfile.mkdtemp()
thrm = (C.c_char_p * 64)(*[os.path.join(td, 't%d' % k).encode() for k in range(64)])
traj = (C.c_char_p * 64)(*[os.path.join(td, 'j%d' % k).encode() for k in range(64)])
def write(rows, xs, bs):
    B.load().nm_append_outputs(64, 256, thrm, traj, rows.ctypes.data_as(B.c_double_p), xs.ctypes.data_as(B.c_double_p), bs.ctypes.data_as(B.c_double_p), 8)
NC = int(os.environ.get('NM_PROBE_CYCLES', '20'))
for mode in ('off', 'snapshot+fetch', 'snapshot+fetch+write', 'off', 'snapshot+fetch', 'off'):
    e.synchronize(); e.timing_reset(); e.stats(reset=True); t0 = time.perf_counter(); snaps = 0; th = None
    for _ in range(NC):
        e.set_step(step); e.run_block(128)
        if mode != 'off':
            e.snapshot(); snaps += 1
        e.adapt(); e.exchange(count=False); step += 1
        if snaps > 1:
            r = e.snapshot_fetch(); snaps -= 1
            if mode.endswith('write'):
                if th: th.join()
                th = threading.Thread(target=write, args=r); th.start()
    while snaps:
        e.snapshot_fetch(); snaps -= 1
    if th: th.join()
    e.synchronize(); dt = time.perf_counter() - t0
    n, ms = e.timing()
    st = e.stats()
    print('%-22s block kernel %.3f ms, cycle %.3f ms, slot block mean %.3f ms, heals %d' % (mode, ms / n, dt / NC * 1e3, (st[:, 4] / st[:, 6] * 1e-5).mean(), e.heals))
e.close()
