#!/bin/bash
# dev: instruction / stall counters of the block kernel for libnm_hip.so and $1 on the same box (two --pmc passes each)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for tag in new old; do
  if [ $tag = old ]; then export NM_HIP_LIB=$R/$1; else unset NM_HIP_LIB; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/$2/$tag.a -- python3 $R/bench.py --config ${CFG:-C2} --no-cpu --steps 4 --warmup 4 > /dev/null 2> $R/gpurun_out/$2/$tag.err
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/$2/$tag.b -- python3 $R/bench.py --config ${CFG:-C2} --no-cpu --steps 4 --warmup 4 > /dev/null 2> $R/gpurun_out/$2/$tag.err
  python3 $R/scripts/collect_pmc.py $R/gpurun_out/$2/$tag.json --skip 2 $R/gpurun_out/$2/$tag.a $R/gpurun_out/$2/$tag.b > /dev/null
done
python3 - <<PY
import json
a=json.load(open('$R/gpurun_out/$2/new.json')); b=json.load(open('$R/gpurun_out/$2/old.json'))
for k in sorted(a):
    if k!='_meta' and k in b: print('%-24s new %14.0f  old %14.0f  ratio %.3f' % (k, a[k]['mean'], b[k]['mean'], a[k]['mean']/max(b[k]['mean'],1)))
PY
