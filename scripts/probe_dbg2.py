import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['NM_HIP_LIB'] = os.path.join(ROOT, 'neuralmelting_amd', 'libnm_hip_exp.so')
from scripts.probe_costs import run
run(0.0, 0.0, 8)
run(0.0, 0.0, 16)
