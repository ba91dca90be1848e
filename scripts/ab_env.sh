#!/bin/bash
# A/B of one environment setting on the SAME box: bench lines without and with "$1" (e.g. NM_PLAIN_GRANULES=1), alternating
for i in 1 2 3; do
  python bench.py --config ${CFG:-C2} --no-cpu --steps ${STEPS:-10} --warmup ${WARM:-5} | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default  %.0f sweeps/s  kernel %.3f ms' % (d['value'], d['roofline']['kernel_avg_ms']))"
  env $1 python bench.py --config ${CFG:-C2} --no-cpu --steps ${STEPS:-10} --warmup ${WARM:-5} | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1  %.0f sweeps/s  kernel %.3f ms' % (d['value'], d['roofline']['kernel_avg_ms']))"
done
