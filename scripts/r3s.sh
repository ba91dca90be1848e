#!/bin/bash
O=$PWD/gpurun_out/r3s; mkdir -p $O
export NM_TESTING=1 NM_ASSUME_CUS=512
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
for v in default b256; do
  if [ $v = default ]; then unset NM_HIP_LIB; else export NM_HIP_LIB=$ROOT/neuralmelting_amd/libnm_hip_b256.so; fi
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_$v -- python3 $ROOT/bench.py --config C2 --steps 10 --warmup 30 --equil 0 --no-cpu > $O/bench_$v.json 2> $O/pmc_$v.err
  python3 $ROOT/scripts/collect_pmc.py $O/pmc_$v.json --config C2 --replicas 64 --mod 128 --commit wip --skip 30 --take 10 $O/pmc_$v > /dev/null
done
python3 - <<PY
import json
for v in ('default','b256'):
    d=json.load(open('$O/pmc_%s.json'%v)); b=json.load(open('$O/bench_%s.json'%v))
    cyc=d['GRBM_GUI_ACTIVE']['mean']/8
    print(v, 'kernel %.3f ms Q=%d'%(b['roofline']['kernel_avg_ms'], b['roofline']['cus_per_replica']), 'VALU active %.3f'%(4*d['SQ_ACTIVE_INST_VALU']['mean']/(1024*cyc)), 'VALU insts %.4g'%d['SQ_INSTS_VALU']['mean'], 'wait %.3f'%(d['SQ_WAIT_ANY']['mean']/d['SQ_WAVE_CYCLES']['mean']), 'busy cycles %.4g wave cycles %.4g'%(d['SQ_BUSY_CYCLES']['mean'], d['SQ_WAVE_CYCLES']['mean']))
PY
