import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.probe_costs import run
run(0.0, 0.0, 1)
run(0.0, 0.0, 8)
