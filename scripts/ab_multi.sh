#!/bin/bash
# A/B/C... of several builds on the SAME box: bench lines of libnm_hip.so and of every library given, alternating, three rounds
for i in 1 2 3; do
  for lib in default "$@"; do
    if [ $lib = default ]; then unset NM_HIP_LIB; else export NM_HIP_LIB=$PWD/$lib; fi
    python bench.py --config ${CFG:-C2} --no-cpu --steps ${STEPS:-10} --warmup ${WARM:-5} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-40s %.0f sweeps/s  kernel %.3f ms' % ('$lib', d['value'], d['roofline']['kernel_avg_ms']))"
  done
done
