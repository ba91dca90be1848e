"""dev probe (experiment build with -DNM_TL_REBUILD: `make -C neuralmelting_amd/csrc exprb`): where does a list rebuild of the 4^3 cluster kernel spend its time?  Stamps of slot 0's
evaluations: 0 entry, 3 conversion issued, 4 past the barrier, 5 tests done, 6 scan + appends done, 7 past the closing reduction.
    NM_TL_TREV=1 python scripts/probe_rebuild.py      (slot 0 = the hottest replica of the first pressure row)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['NM_HIP_LIB'] = os.path.join(ROOT, 'neuralmelting_amd', 'libnm_hip_exp_rb.so')
import numpy as np
import neuralmelting_amd as nm
from neuralmelting_amd import lattice, _lib
P = np.linspace(1, 8, 8, dtype=np.float32); T = np.linspace(.25, 2.5, 8, dtype=np.float32)
if os.environ.get('NM_TL_TREV'):
    T = T[::-1].copy()
x, v, box, d = lattice.init_states(4, P, T, 0.03125, 0.03125)
e = nm.Engine(256, P, T)
e.set_state(x, v, box, d)
for s in range(32):
    e.set_step(s); e.run_block(128); e.adapt(); e.exchange(count=False)
e.synchronize()
L = _lib.load()
L.nm_tline_get.argtypes = [C.c_void_p, C.c_void_p]
buf = np.zeros((8, 8, 512, 8), dtype=np.uint64)
L.nm_tline_get(e.h, buf.ctypes.data)
Q = e.cus_per_replica
t = buf[:Q].astype(np.int64)
# evaluations that rebuilt: stamp 7 present and later than stamp 0 of the same evaluation
rb = [k for k in range(20, 500) if t[0, 0, k, 7] > t[0, 0, k, 0] > 0 and t[0, 0, k, 3] > t[0, 0, k, 0]]
print('evaluations with a rebuild: %d of 480' % len(rb))
names = [('entry -> conversion issued', 0, 3), ('-> past the barrier', 3, 4), ('-> tests done', 4, 5), ('-> scan + appends done', 5, 6), ('-> x0 copy + closing reduction', 6, 7), ('whole rebuild', 0, 7)]
for nme, a, b in names:
    dd = np.array([[(t[q, w, k, b] - t[q, w, k, a]) * 10.0 for k in rb] for q in range(Q) for w in range(8)])
    print('  %-34s median over waves and rebuilds %6.0f ns   slowest wave (median over rebuilds) %6.0f ns' % (nme, np.median(dd), np.median(dd.max(0))))
e.close()
