#!/bin/bash
# round-3 evidence run: tests, diagnostic self-checks, profiles of every preset (equilibrated launches), bench lines
O=gpurun_out/r3j; mkdir -p $O
C=$(cat $O/../commit.txt 2>/dev/null || echo wip)
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
NM_HIP_LIB=$PWD/neuralmelting_amd/libnm_hip_prof.so timeout -k 10 600 python scripts/check_bounds.py > $O/check_bounds.txt 2>&1; echo "bounds rc=$?"; tail -1 $O/check_bounds.txt
for cfg in "C2 64" "C3 32" "C4 64" "C5 128" "runsh 1024"; do
  set -- $cfg
  timeout -k 10 500 bash scripts/profile_round.sh $1 $2 $C > $O/prof_$1.log 2>&1; echo "profile $1 rc=$?"
done
timeout -k 10 500 bash scripts/profile_round.sh C2 64 $C "--iterative" C2_iter > $O/prof_C2_iter.log 2>&1; echo "profile C2_iter rc=$?"
# the hand-over's HBM-side traffic with write-through granules everywhere (no plain stores inside one XCD)
cd /tmp && export TMPDIR=/tmp
NM_PLAIN_GRANULES=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OLDPWD/$O/pmc_write_wt -- python3 $OLDPWD/bench.py --config C2 --steps 10 --warmup 30 --equil 0 --no-cpu > /dev/null 2> $OLDPWD/$O/pmc_write_wt.err
NM_PLAIN_GRANULES=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OLDPWD/$O/pmc_fetch_wt -- python3 $OLDPWD/bench.py --config C2 --steps 10 --warmup 30 --equil 0 --no-cpu > /dev/null 2> $OLDPWD/$O/pmc_fetch_wt.err
cd $OLDPWD
python3 scripts/collect_pmc.py $O/pmc_C2_write_through.json --config C2 --replicas 64 --mod 128 --commit $C --skip 30 --take 10 --regime "equilibrated, NM_PLAIN_GRANULES=0" $O/pmc_write_wt $O/pmc_fetch_wt > /dev/null
