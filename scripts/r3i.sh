#!/bin/bash
O=gpurun_out/r3i; mkdir -p $O
WARM=5 STEPS=20 bash scripts/ab_multi.sh neuralmelting_amd/libnm_hip_ab1.so neuralmelting_amd/libnm_hip_ab4.so 2>&1 | tee $O/ab.txt
