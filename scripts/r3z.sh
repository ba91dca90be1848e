#!/bin/bash
# full GPU tests + bounds/list self-check (diagnostic build) + A/B of the working tree's library against neuralmelting_amd/libnm_hip_prev.so
O=gpurun_out/r3z; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
[ -f neuralmelting_amd/libnm_hip_prof.so ] && { NM_HIP_LIB=$PWD/neuralmelting_amd/libnm_hip_prof.so timeout -k 10 600 python scripts/check_bounds.py > $O/check_bounds.txt 2>&1; echo "bounds rc=$?"; tail -1 $O/check_bounds.txt; }
WARM=5 STEPS=20 bash scripts/ab_multi.sh neuralmelting_amd/libnm_hip_prev.so 2>&1 | tee $O/ab_C2.txt
for cfg in ${CFGS:-C3}; do CFG=$cfg WARM=8 STEPS=10 REPS=2 bash scripts/ab_multi.sh neuralmelting_amd/libnm_hip_prev.so 2>&1 | tee $O/ab_$cfg.txt; done
