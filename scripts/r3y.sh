#!/bin/bash
O=gpurun_out/r3y; mkdir -p $O
for lib in default prev; do
  if [ $lib = default ]; then unset NM_HIP_LIB; else export NM_HIP_LIB=$PWD/neuralmelting_amd/libnm_hip_prev.so; fi
  timeout -k 10 500 python - > $O/repeat_$lib.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
os.environ['NM_TESTING'] = '1'
import numpy as np
from helpers import OracleLoop, grids
from oracle import oracle as O
import neuralmelting_amd as nm
O.build(); O.lib()
bad = 0
for sz, cus in ((6, 4), (6, 2), (5, 4), (6, 8), (8, 2)):
    os.environ['NM_CUS_PER_REPLICA'] = str(cus)
    P, T = grids(1, 2)
    kw = dict(bulk=True, ppos=0.3, pvol=0.2)
    loop = OracleLoop(O, sz, P, T, **kw)
    x0, v0, b0, d0 = loop.x.copy(), loop.v.copy(), loop.box.copy(), loop.d.copy()
    loop.run_block(6, 0)
    ro = loop.rows()
    for rep in range(12):
        e = nm.Engine(4 * sz ** 3, P, T, row0=loop.row0, nrows=loop.nrows, seed=loop.seed, **kw)
        e.set_state(x0, v0, b0, d0)
        e.run_block(6)
        rows = e.thermo()
        err = np.abs(rows[:, :5] / ro[:, :5] - 1).max()
        x, v, box, d = e.get_state()
        e.close()
        flag = 'BAD' if err > 1e-6 else 'ok'
        if err > 1e-6: bad += 1
        print(sz, cus, rep, flag, '%.3g' % err, 'max |dv| %.3g' % np.abs(v - loop.v).max(), flush=True)
print('bad', bad)
PY
  tail -1 $O/repeat_$lib.txt; grep -c BAD $O/repeat_$lib.txt
done
