#!/bin/bash
# dev: LDS bank-conflict share of the block kernel for a few supercell sizes (4 x 32 grid each), one --pmc pass per size
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for sz in "$@"; do
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/lds/sz$sz -- python3 $R/bench.py --sz $sz --rows 4 --tn 32 --no-cpu --steps 3 --warmup 3 > /dev/null 2> $R/gpurun_out/lds/sz$sz.err
  python3 $R/scripts/collect_pmc.py $R/gpurun_out/lds/sz$sz.json --skip 2 $R/gpurun_out/lds/sz$sz > /dev/null
  python3 -c "
import json; d=json.load(open('$R/gpurun_out/lds/sz$sz.json'))
print('sz $sz: bank conflict / lds active = %.2f ; conflict share of CU cycles = %.3f' % (d['SQ_LDS_BANK_CONFLICT']['mean']/d['SQ_ACTIVE_INST_LDS']['mean'], d['SQ_LDS_BANK_CONFLICT']['mean']/(256*d['GRBM_GUI_ACTIVE']['mean']/8)))"
  rm -rf $R/gpurun_out/lds/sz$sz
done
