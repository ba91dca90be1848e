#!/bin/bash
O=gpurun_out/r3d; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
NM_HIP_LIB=$PWD/neuralmelting_amd/libnm_hip_prof.so timeout -k 10 600 python scripts/check_bounds.py > $O/bounds.txt 2>&1; echo "bounds rc=$?"; tail -16 $O/bounds.txt
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
for sk in 0.4 0.3 0.35 0.45 0.4; do b C2_skin$sk NM_SKIN=$sk "--config C2 --warmup 5 --steps 20"; done
for sk in 0.6 0.45 0.75; do b C5_skin$sk NM_SKIN=$sk "--config C5 --warmup 8 --steps 6"; done
b C4 NM_X=0 "--config C4 --warmup 8 --steps 10"
for sk in 0.6 1.0; do b C4_skin$sk NM_SKIN_AL=$sk "--config C4 --warmup 8 --steps 10"; done
b C3 NM_X=0 "--config C3 --warmup 8 --steps 10"
b runsh NM_X=0 "--config runsh --warmup 8 --steps 4"
b C2_iter NM_X=0 "--config C2 --warmup 5 --steps 10 --iterative"
# one rank's leg of the metric's own 8x8 grid at N = 2 / 4 / 8 GPUs (strong scaling): 4 / 2 / 1 pressure rows
for r in 4 2 1; do b C2_rows$r NM_X=0 "--rows $r --tn 8 --warmup 5 --steps 20"; done
b C2_rows4_Q4 NM_CUS_PER_REPLICA=4 "--rows 4 --tn 8 --warmup 5 --steps 20"
timeout -k 10 200 python scripts/probe_balance.py C2 30 10 4 > $O/balance_C2_rows4.txt 2>&1; head -2 $O/balance_C2_rows4.txt
