#!/bin/bash
# A/B of two builds on the SAME box (boxes differ by several per cent): bench lines of libnm_hip.so and of $1, alternating
for i in 1 2 3; do
  python bench.py --config ${CFG:-C2} --no-cpu --steps ${STEPS:-10} --warmup ${WARM:-5} | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default  %.0f sweeps/s  kernel %.3f ms' % (d['value'], d['roofline']['kernel_avg_ms']))"
  NM_HIP_LIB=$PWD/$1 python bench.py --config ${CFG:-C2} --no-cpu --steps ${STEPS:-10} --warmup ${WARM:-5} | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant  %.0f sweeps/s  kernel %.3f ms' % (d['value'], d['roofline']['kernel_avg_ms']))"
done
