#!/bin/bash
O=gpurun_out/r3c; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for c in C2 C5 C3 runsh C4; do
  timeout -k 10 400 python bench.py --config $c --warmup 8 --steps 10 --no-cpu > $O/bench_$c.json 2> $O/bench_$c.err; echo "bench $c rc=$?"
done
timeout -k 10 300 python bench.py --warmup 5 --steps 10 --iterative --no-cpu > $O/bench_C2_iter.json 2> $O/bench_C2_iter.err
timeout -k 10 300 python scripts/probe_balance.py C5 40 6 > $O/balance_C5_eq.txt 2>&1
timeout -k 10 200 python scripts/probe_balance.py C2 30 10 > $O/balance_C2_eq.txt 2>&1
python - <<'PY'
import json
for c in ('C2','C5','C3','runsh','C4','C2_iter'):
    try:
        d=json.load(open('gpurun_out/r3c/bench_%s.json'%c))
        print(c, 'window %.0f (%.2f ms)  sustained %.0f (%.2f ms, rebuilds %.2f, frac %.4f) slot mean/max %.2f/%.2f' % (d['window']['value'], d['window']['kernel_avg_ms'], d['sustained']['value'], d['sustained']['kernel_avg_ms'], d['sustained']['list_rebuilds_per_sweep'], d['sustained']['frac'], d['sustained']['slot_block_ms_mean'], d['sustained']['slot_block_ms_max']))
    except Exception as e: print(c, 'failed', e)
PY
