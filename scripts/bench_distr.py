"""throughput of nm_distr_histograms (row f-2) next to the numpy restatement of calculate_rdf / calculate_cdf"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralmelting_amd import distr
from oracle import distr_oracle as D

ns, n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 256
rng = np.random.default_rng(3)
box = (6.0 + 0.3 * rng.random(ns)).astype(np.float32)
pos = (rng.random((ns, n, 3)) * box[:, None, None]).astype(np.float32)
natoms = np.full(ns, n, np.uint16)
nrho, dni, r, dn, rv = distr.calculate_spatial(natoms, box, 64, 16)
distr.histograms(natoms[:8], box[:8], pos[:8], r, rv)
for rep in range(3):
    t = time.time(); rdf, cdf = distr.histograms(natoms, box, pos, r, rv); dt = time.time() - t
    print('GPU call (H2D + kernel + D2H): %d samples x %d atoms in %.3f s = %.0f samples/s, %.2f G pair-images/s'
          % (ns, n, dt, ns / dt, ns * 27 * n * n / dt / 1e9))
m = 8
t = time.time()
for s in range(m):
    D.calculate_rdf(n, box[s], pos[s], r); D.calculate_cdf(n, box[s], pos[s], rv)
dt = time.time() - t
print('numpy restatement: %.1f samples/s (one core)' % (m / dt))
