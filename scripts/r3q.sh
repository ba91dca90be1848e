#!/bin/bash
O=gpurun_out/r3q; mkdir -p $O
NM_HIP_LIB=$PWD/neuralmelting_amd/libnm_hip_b256.so timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_properties_gpu.py -m gpu -x -q -k "not large_cells and not fall" > $O/pytest_b256.log 2>&1; echo "pytest b256 rc=$?"; tail -5 $O/pytest_b256.log
WARM=5 STEPS=20 bash scripts/ab_multi.sh neuralmelting_amd/libnm_hip_b256.so 2>&1 | tee $O/ab_C2.txt
NM_HIP_LIB=$PWD/neuralmelting_amd/libnm_hip_b256.so timeout -k 10 200 python scripts/probe_balance.py C2 30 10 2>&1 | head -3
