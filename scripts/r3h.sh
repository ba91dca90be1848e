#!/bin/bash
O=gpurun_out/r3h; mkdir -p $O
bash scripts/profile_round.sh C5 128 wip > $O/prof_C5.log 2>&1; tail -3 $O/prof_C5.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/prof_C5/r03_pmc_block_kernel_C5.json'))
for k,v in d.items():
    if k!='_meta': print(k, '%.4g'%v['mean'])
cyc=d['GRBM_GUI_ACTIVE']['mean']/8
print('VALU active share', 4*d['SQ_ACTIVE_INST_VALU']['mean']/(1024*cyc))
print('LDS conflict / LDS active', d['SQ_LDS_BANK_CONFLICT']['mean']/d['SQ_ACTIVE_INST_LDS']['mean'])
print('LDS conflict share of CU cycles', d['SQ_LDS_BANK_CONFLICT']['mean']/(256*cyc))
print('WAIT_ANY/WAVE_CYCLES', d['SQ_WAIT_ANY']['mean']/d['SQ_WAVE_CYCLES']['mean'])
print('traffic GB', (2*d['FETCH_SIZE']['mean']+d['WRITE_SIZE']['mean'])*1024/1e9)
PY
