#!/bin/bash
# A/B/C... of several builds on the SAME box (boxes differ by several per cent): bench lines of libnm_hip.so and of every library
# given, alternating, three rounds; window and sustained rate of each
for i in $(seq 1 ${REPS:-3}); do
  for lib in default "$@"; do
    if [ $lib = default ]; then unset NM_HIP_LIB; else export NM_HIP_LIB=$PWD/$lib; fi
    python bench.py --config ${CFG:-C2} --no-cpu --steps ${STEPS:-10} --warmup ${WARM:-5} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-44s window %8.0f (%.3f ms)  sustained %8.0f (%.3f ms) Q=%d' % ('$lib', d['window']['value'], d['window']['kernel_avg_ms'], d['sustained']['value'], d['sustained']['kernel_avg_ms'], d['roofline']['cus_per_replica']))"
  done
done
