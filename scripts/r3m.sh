#!/bin/bash
O=gpurun_out/r3m; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 300 python bench.py --rows 2 --tn 2 --no-cpu > $O/bench_C1.json 2> $O/bench_C1.err; python -c "
import json; d=json.load(open('$O/bench_C1.json')); print('C1 sustained %.0f window %.0f kernel %.2f one-xcd %s Q=%d' % (d['value'], d['window']['value'], d['roofline']['kernel_avg_ms'], d['roofline']['clusters_on_one_xcd'], d['roofline']['cus_per_replica']))"
timeout -k 10 300 python bench.py --iterative --cpu-seconds 4 > $O/bench_C2_iter.json 2> $O/bench_C2_iter.err; python -c "
import json; d=json.load(open('$O/bench_C2_iter.json')); print('iter value %.0f kernel %.2f rebuilds %.2f cpu %s' % (d['value'], d['roofline']['kernel_avg_ms'], d['roofline']['list_rebuilds_per_sweep'], d.get('cpu_baseline',{}).get('value')))"
timeout -k 10 300 python bench.py --iterative --warmup 3 --steps 5 --no-cpu > $O/bench_C2_iter_early.json 2> $O/bench_C2_iter_early.err; python -c "
import json; d=json.load(open('$O/bench_C2_iter_early.json')); print('iter early value %.0f kernel %.2f rebuilds %.2f' % (d['value'], d['roofline']['kernel_avg_ms'], d['roofline']['list_rebuilds_per_sweep']))"
