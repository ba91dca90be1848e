"""Would one stream per pressure row pay?  The exchange never leaves a row (remcmc:782-798), so rows need not wait for one another.
This probe runs the C2 grid (a) as ONE context = one launch per cycle over all 64 replicas and (b) as EIGHT contexts of one row each
on the same GPU (each with its own stream, NM_CUS_PER_REPLICA=4 so that the eight 32-workgroup grids fill the chip together), all
cycles enqueued without a host wait in between, and prints the two rates.  No kernel change involved.

    NM_CUS_PER_REPLICA=4 python scripts/probe_rowstreams.py [warm cycles]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import neuralmelting_amd as nm
from neuralmelting_amd import lattice


def run(engines, warm, cycles, mod):
    for s in range(warm):
        for e in engines:
            e.set_step(s); e.run_block(mod); e.adapt(); e.exchange(count=False)
    for e in engines:
        e.synchronize()
    t0 = time.perf_counter()
    for s in range(warm, warm + cycles):
        for e in engines:
            e.set_step(s); e.run_block(mod); e.adapt(); e.exchange(count=False)
    for e in engines:
        e.synchronize()
    dt = time.perf_counter() - t0
    ns = sum(e.nslots for e in engines)
    return ns * mod * cycles / dt, dt / cycles * 1e3


def main(warm=30, cycles=20, mod=128):
    sz, npn, tn = 4, 8, 8
    P = np.linspace(1, 8, npn, dtype=np.float32); T = np.linspace(.25, 2.5, tn, dtype=np.float32)
    x, v, box, d = lattice.init_states(sz, P, T, 0.03125, 0.03125)
    one = nm.Engine(256, P, T)
    one.set_state(x, v, box, d)
    r1, ms1 = run([one], warm, cycles, mod)
    th1 = one.thermo()
    print('one context  (Q = %d): %.0f sweeps/s, %.3f ms per cycle' % (one.cus_per_replica, r1, ms1))
    one.close()
    many = []
    for r in range(npn):
        e = nm.Engine(256, P, T, row0=r, nrows=1)
        e.set_state(x[r * tn:(r + 1) * tn], v[r * tn:(r + 1) * tn], box[r * tn:(r + 1) * tn], d[r * tn:(r + 1) * tn])
        many.append(e)
    r8, ms8 = run(many, warm, cycles, mod)
    th8 = np.concatenate([e.thermo() for e in many])
    print('eight contexts (Q = %s): %.0f sweeps/s, %.3f ms per cycle  (%+.1f %%)' % (many[0].cus_per_replica, r8, ms8, 100 * (r8 / r1 - 1)))
    print('results identical:', bool(np.array_equal(th1, th8)))
    for e in many:
        print('  row %d: note=%r' % (e.row0, e.lib.nm_create_note(e.h).decode()))
        e.close()


if __name__ == '__main__':
    a = [int(v) for v in sys.argv[1:]]
    main(*a)
