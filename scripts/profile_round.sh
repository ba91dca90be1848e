#!/bin/bash
# rocprofv3 evidence for one bench preset, run on the GPU box from the repo root:
#   scripts/profile_round.sh C2 64 <commit>      -> gpurun_out/prof_C2/{stats,pmc_*}, profiles-ready summaries next to them
# kernel stats (one run) and three separate counter passes (never --pmc together with a trace flag)
set -e
CFG=$1; NS=$2; COMMIT=$3; STEPS=${4:-10}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$CFG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --config $CFG --steps $STEPS --warmup 8 --no-cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/bench_stats.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
          --output-format csv -d $OUT/pmc_sq -- $B > /dev/null 2> $OUT/pmc_sq.err
cd $ROOT
python3 scripts/collect_pmc.py $OUT/r02_pmc_block_kernel_$CFG.json --config $CFG --replicas $NS --mod 128 --commit $COMMIT --skip 8 \
        $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq > /dev/null
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/r02_kernel_stats_$CFG.csv
head -5 $OUT/r02_kernel_stats_$CFG.csv
