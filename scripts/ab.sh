#!/bin/bash
# A/B/C... on the SAME box (boxes differ by several per cent): bench lines of the shipped libnm_hip.so ("default") and of every variant
# given, alternating, REPS rounds; window and sustained rate of each.  A variant is either a library path relative to the repo root
# (built with `make -C neuralmelting_amd/csrc ab<k>` or any -D flag) or an environment setting VAR=VALUE (e.g. NM_SKIN=0.37).
#   CFG=C2 REPS=3 STEPS=10 WARM=5 EXTRA="--rows 4" scripts/ab.sh neuralmelting_amd/libnm_hip_ab3.so NM_PLAIN_GRANULES=0
for i in $(seq 1 ${REPS:-3}); do
  for var in default "$@"; do
    unset NM_HIP_LIB; pre=""
    case "$var" in
      default) ;;
      *=*) pre="$var" ;;
      *) export NM_HIP_LIB=$PWD/$var ;;
    esac
    env $pre python bench.py --config ${CFG:-C2} --no-cpu --steps ${STEPS:-10} --warmup ${WARM:-5} $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-44s window %8.0f (%.3f ms)  sustained %8.0f (%.3f ms) Q=%d' % ('$var', d['window']['value'], d['window']['kernel_avg_ms'], d['sustained']['value'], d['sustained']['kernel_avg_ms'], d['roofline']['cus_per_replica']))"
  done
done
