#!/bin/bash
# rocprofv3 evidence for one bench preset, run on the GPU box from the repo root:
#   scripts/profile_round.sh C2 64 <commit> [extra bench flags]   -> gpurun_out/prof_C2/{stats,pmc_*}, profiles-ready summaries next to them
# kernel stats (one run) and three separate counter passes (never --pmc together with a trace flag).  The profiled launches are the
# EQUILIBRATED ones: 30 warm-up cycles from the lattice start (step sizes adapted, HMC accepting about half its trajectories), then
# the 10 timed cycles, no second phase (--equil 0): the regime of the bench line's `value` (its `sustained` object).  Where the bench runs its timed
# region as ONE launch of nm_cycles_kernel (nm_run_cycles) that launch is what the counters describe; elsewhere the 10 nm_block_kernel launches.
set -e
CFG=$1; NS=$2; COMMIT=$3; EXTRA=$4; TAG=${5:-$CFG}; RND=${ROUND:-r04}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --config $CFG --steps 10 --warmup 30 --equil 0 --no-cpu $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/bench_stats.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
          --output-format csv -d $OUT/pmc_sq -- $B > /dev/null 2> $OUT/pmc_sq.err
cd $ROOT
python3 scripts/collect_pmc.py $OUT/${RND}_pmc_block_kernel_$TAG.json --config $CFG --replicas $NS --mod 128 --commit $COMMIT --skip 30 --take 10 --kernel auto --cycles 10 \
        --regime "equilibrated: cycles 30-39 after the lattice start" $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq > /dev/null
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/${RND}_kernel_stats_$TAG.csv
head -5 $OUT/${RND}_kernel_stats_$TAG.csv
