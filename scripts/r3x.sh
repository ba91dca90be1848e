#!/bin/bash
O=gpurun_out/r3x; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_eam.py tests/test_shares_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
WARM=5 STEPS=20 bash scripts/ab_multi.sh neuralmelting_amd/libnm_hip_prev.so 2>&1 | tee $O/ab_C2.txt
CFG=C4 WARM=8 STEPS=10 bash scripts/ab_multi.sh neuralmelting_amd/libnm_hip_prev.so 2>&1 | tee $O/ab_C4.txt
