#!/bin/bash
O=gpurun_out/r3n; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
WARM=5 STEPS=20 bash scripts/ab_multi.sh neuralmelting_amd/libnm_hip_ab5.so 2>&1 | tee $O/ab_C2.txt
CFG=C4 WARM=8 STEPS=10 bash scripts/ab_multi.sh neuralmelting_amd/libnm_hip_ab5.so 2>&1 | tee $O/ab_C4.txt
timeout -k 10 300 python bench.py --iterative --no-cpu > $O/bench_C2_iter.json 2> $O/bench_C2_iter.err; python -c "
import json; d=json.load(open('$O/bench_C2_iter.json')); print('iter value %.0f kernel %.2f rebuilds %.2f warm %d steps %d' % (d['value'], d['roofline']['kernel_avg_ms'], d['roofline']['list_rebuilds_per_sweep'], d['warmup'], d['steps']))"
timeout -k 10 300 python scripts/iter_watch.py 30 > $O/iter_watch.txt 2>&1; tail -4 $O/iter_watch.txt
