#!/bin/bash
O=gpurun_out/r3p; mkdir -p $O
WARM=5 STEPS=20 bash scripts/ab_multi.sh neuralmelting_amd/libnm_hip_ps0.so neuralmelting_amd/libnm_hip_ps2.so neuralmelting_amd/libnm_hip_ps4.so 2>&1 | tee $O/ab_C2.txt
