#!/bin/bash
# a round's evidence at one commit ($1), in two calls that each fit a 20-minute GPU slot:  ROUND=r04 scripts/evidence.sh <commit> a|b
#   a: tests, smoke, diagnostic self-checks, rocprofv3 profiles of every preset (equilibrated launches)
#   b: the bench lines (driver's flags, every preset, C1, iterative, row scan, recording on), section shares
C=$1; PART=${2:-a}
RND=${ROUND:-r04}; export ROUND=$RND; O=gpurun_out/final_$RND; mkdir -p $O
if [ $PART = a ]; then
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
NM_HIP_LIB=$PWD/neuralmelting_amd/libnm_hip_prof.so timeout -k 10 400 python scripts/check_bounds.py > $O/check_bounds.txt 2>&1; echo "bounds rc=$?"; tail -1 $O/check_bounds.txt
for cfg in "C2 64" "C3 32" "C4 64" "C5 128" "runsh 1024"; do
  set -- $cfg
  timeout -k 10 300 bash scripts/profile_round.sh $1 $2 $C > $O/prof_$1.log 2>&1; echo "profile $1 rc=$?"
done
else
timeout -k 10 300 python bench.py --warmup 5 --steps 20 > $O/bench_driver.json 2> $O/bench_driver.err; echo "driver-like rc=$?"
for c in C2 C3 C4 C5 runsh; do timeout -k 10 400 python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; echo "bench $c rc=$?"; done
NM_OVERSUBSCRIBE=1 timeout -k 10 400 python bench.py --config C5 --no-cpu > $O/bench_C5x2.json 2> $O/bench_C5x2.err; echo "bench C5x2 rc=$?"
timeout -k 10 300 python bench.py --rows 2 --tn 2 > $O/bench_C1.json 2> $O/bench_C1.err; echo "bench C1 rc=$?"
timeout -k 10 300 python bench.py --iterative --cpu-seconds 4 > $O/bench_C2_iter.json 2> $O/bench_C2_iter.err; echo "bench iter rc=$?"
for r in 4 2 1; do timeout -k 10 300 python bench.py --rows $r --tn 8 --warmup 5 --steps 20 --no-cpu > $O/bench_C2_rows$r.json 2> $O/bench_C2_rows$r.err; done
timeout -k 10 300 python bench.py --record --no-cpu > $O/bench_C2_record.json 2> $O/bench_C2_record.err; echo "bench C2 record rc=$?"
timeout -k 10 400 python bench.py --config C5 --record --no-cpu > $O/bench_C5_record.json 2> $O/bench_C5_record.err; echo "bench C5 record rc=$?"
timeout -k 10 200 python scripts/probe_sections.py 4 8 8 128 10 30 > $O/sections_C2_eq.txt 2>&1
timeout -k 10 200 python scripts/probe_sections.py 4 4 8 128 10 30 > $O/sections_C2_Q8_32replicas.txt 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/final_%s/bench_*.json' % __import__('os').environ.get('ROUND', 'r04'))):
    try:
        d=json.load(open(f))
        cb=d.get('cpu_baseline') or {}
        print('%-10s value %9.0f window %9.0f kernel %.2f/%.2f ms frac %.4f/%.4f rb %.2f cpu %s/%s Q=%d prof %s' % (f.split('bench_')[1][:-5], d['value'], d['window']['value'], d['roofline']['kernel_avg_ms'], d['window']['kernel_avg_ms'], d['roofline']['frac'], d['window']['frac'], d['roofline']['list_rebuilds_per_sweep'], '%.0f'%cb['value'] if cb else '-', '%.0f'%cb['single_thread']['value'] if cb else '-', d['roofline']['cus_per_replica'], d['roofline']['profile_commit']))
    except Exception as e: print(f, 'failed', e)
PY
fi
