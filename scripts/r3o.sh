#!/bin/bash
O=gpurun_out/r3o; mkdir -p $O
WARM=5 STEPS=20 bash scripts/ab_multi.sh neuralmelting_amd/libnm_hip_ab7.so neuralmelting_amd/libnm_hip_relocc.so neuralmelting_amd/libnm_hip_trk.so 2>&1 | tee $O/ab_C2.txt
