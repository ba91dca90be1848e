#!/bin/bash
O=gpurun_out/r3f; mkdir -p $O
NM_HIP_LIB=$PWD/neuralmelting_amd/libnm_hip_prof.so timeout -k 10 600 python scripts/check_bounds.py > $O/bounds.txt 2>&1; echo "bounds rc=$?"; tail -16 $O/bounds.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
b() { # name, extra env, args
  env $2 timeout -k 10 400 python bench.py $3 --no-cpu > $O/bench_$1.json 2> $O/bench_$1.err
  python - "$O/bench_$1.json" "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print('%-14s window %8.0f (%6.2f ms)  sustained %8.0f (%6.2f ms, rebuilds %.2f, frac %.4f) slot mean/max %.2f/%.2f Q=%d' % (sys.argv[2], d['window']['value'], d['window']['kernel_avg_ms'], d['sustained']['value'], d['sustained']['kernel_avg_ms'], d['sustained']['list_rebuilds_per_sweep'], d['sustained']['frac'], d['sustained']['slot_block_ms_mean'], d['sustained']['slot_block_ms_max'], d['roofline']['cus_per_replica']))
except Exception as e: print(sys.argv[2], 'failed', e)
PY
}
b C2 NM_X=0 "--config C2 --warmup 5 --steps 20"
b C5 NM_X=0 "--config C5 --warmup 8 --steps 6"
b C5_skin0.5 NM_SKIN=0.5 "--config C5 --warmup 8 --steps 6"
b C3 NM_X=0 "--config C3 --warmup 8 --steps 10"
b runsh NM_X=0 "--config runsh --warmup 8 --steps 4"
b C4 NM_X=0 "--config C4 --warmup 8 --steps 10"
b C2_iter NM_X=0 "--config C2 --warmup 5 --steps 10 --iterative"
timeout -k 10 300 python scripts/probe_balance.py C5 40 6 > $O/balance_C5_eq.txt 2>&1; head -1 $O/balance_C5_eq.txt
