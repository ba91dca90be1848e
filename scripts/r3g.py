import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
os.environ['NM_HIP_LIB'] = os.path.join(os.getcwd(), 'neuralmelting_amd', 'libnm_hip_prof.so')
os.environ['NM_CUS_PER_REPLICA'] = '2'
import numpy as np
import neuralmelting_amd as nm
from neuralmelting_amd import lattice, _lib
L = _lib.load()
Tg = np.linspace(256.0, 2560.0, 8, dtype=np.float32); Pg = np.linspace(1.0, 8.0, 2, dtype=np.float32)
x, v, box, d = lattice.init_states(4, Pg, Tg, 0.03125, 0.03125, el='Al')
e = nm.Engine(256, Pg, Tg, element='Al', bulk=False)
e.set_state(x, v, box, d)
for step in range(10):
    e.set_step(step); e.run_block(64)
    m = C.c_uint(0); L.nm_prof_list_miss(e.h, C.byref(m))
    rows = e.thermo()
    print('cycle', step, 'misses', m.value, 'max |pe|/N %.3g' % (np.abs(rows[:, 1]).max() / 256), 'dx max %.3f' % rows[:, 5].max(), flush=True)
    if m.value:
        info = np.zeros(16); L.nm_prof_miss_info(e.h, info.ctypes.data_as(C.POINTER(C.c_double)))
        print('   slot %d atom %d lacks atom %d; exact %d listed %d row %d L %.6f radius %.3f' % tuple(info[:8]))
        xi, xj, Lb = info[8:11], info[11:14], info[6]
        dd = xi - xj; dd -= Lb * np.rint(dd / Lb)
        print('   x_i', xi, 'x_j', xj, 'sep %.9f' % np.sqrt((dd * dd).sum()), 'fixed i %x j %x' % (int(info[14]), int(info[15])))
    e.adapt(); e.exchange(count=False)
e.close()
