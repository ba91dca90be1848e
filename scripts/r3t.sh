#!/bin/bash
O=gpurun_out/r3t; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for ord in 0 1 0 1; do
NM_LAUNCH_ORDER=$ord timeout -k 10 400 python bench.py --config runsh --warmup 8 --steps 6 --no-cpu > $O/bench_runsh_$ord.json 2> $O/bench_runsh_$ord.err; python -c "
import json; d=json.load(open('$O/bench_runsh_$ord.json')); print('order $ord: window %.0f (%.1f ms) sustained %.0f (%.1f ms) slot mean/max %.1f/%.1f' % (d['window']['value'], d['window']['kernel_avg_ms'], d['sustained']['value'], d['sustained']['kernel_avg_ms'], d['sustained']['slot_block_ms_mean'], d['sustained']['slot_block_ms_max']))"
done
