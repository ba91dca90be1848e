"""dev probe: per-move cost of the 6^3 / 8^3 kernels by move mix"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.probe_costs import run
sz = int(sys.argv[1]); rows = int(sys.argv[2]); tn = int(sys.argv[3])
for (pp, pv, ns) in ((1.0, 0.0, 8), (0.0, 0.0, 1), (0.0, 0.0, 8), (0.125, 0.125, 8)):
    run(pp, pv, ns, mod=32, cycles=2, sz=sz, rows=rows, tn=tn, warm=2)
