#!/bin/bash
O=gpurun_out/r3r; mkdir -p $O
export NM_TESTING=1 NM_ASSUME_CUS=512
NM_HIP_LIB=$PWD/neuralmelting_amd/libnm_hip_b256.so timeout -k 10 200 python - <<'PY'
import numpy as np, neuralmelting_amd as nm
P=np.linspace(1,8,8,dtype=np.float32); T=np.linspace(.25,2.5,8,dtype=np.float32)
e=nm.Engine(256,P,T); print('Q', e.cus_per_replica, 'note', e.note()); e.close()
PY
WARM=5 STEPS=20 bash scripts/ab_multi.sh neuralmelting_amd/libnm_hip_b256.so 2>&1 | tee $O/ab_C2.txt
