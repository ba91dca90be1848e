#!/bin/bash
O=gpurun_out/final3; mkdir -p $O
timeout -k 10 300 python bench.py --warmup 5 --steps 20 > $O/bench_driver.json 2> $O/bench_driver.err; echo "driver-like rc=$?"
for c in C2 C3 C4 C5 runsh; do timeout -k 10 500 python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; echo "bench $c rc=$?"; done
timeout -k 10 300 python bench.py --rows 2 --tn 2 > $O/bench_C1.json 2> $O/bench_C1.err; echo "bench C1 rc=$?"
timeout -k 10 300 python bench.py --iterative --equil 16 --cpu-seconds 4 > $O/bench_C2_iter.json 2> $O/bench_C2_iter.err; echo "bench iter rc=$?"; tail -1 $O/bench_C2_iter.err
for r in 4 2 1; do timeout -k 10 300 python bench.py --rows $r --tn 8 --warmup 5 --steps 20 --no-cpu > $O/bench_C2_rows$r.json 2> $O/bench_C2_rows$r.err; done
timeout -k 10 200 python scripts/probe_sections.py 4 8 8 128 10 30 > $O/sections_C2_eq.txt 2>&1
NM_PROBE_ITER=1 timeout -k 10 200 python scripts/probe_sections.py 4 8 8 128 6 3 > $O/sections_C2_iter_cycles3-8.txt 2>&1
NM_PROBE_ITER=1 timeout -k 10 200 python scripts/probe_sections.py 4 8 8 128 6 16 > $O/sections_C2_iter_cycles16-21.txt 2>&1
timeout -k 10 200 python scripts/probe_sections.py 8 4 32 128 4 34 > $O/sections_C5_eq.txt 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/final3/bench_*.json')):
    try:
        d=json.load(open(f))
        cb=d.get('cpu_baseline') or {}
        print('%-22s value %9.0f window %9.0f kernel %.2f ms frac %.4f rebuilds %.2f cpu %s/%s Q=%d' % (f.split('bench_')[1][:-5], d['value'], d['window']['value'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'], d['roofline']['list_rebuilds_per_sweep'], '%.0f'%cb['value'] if cb else '-', '%.0f'%cb['single_thread']['value'] if cb else '-', d['roofline']['cus_per_replica']))
    except Exception as e: print(f, 'failed', e)
PY
