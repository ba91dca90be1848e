#!/bin/bash
# skin scan on one box: sustained / window rate of a preset for several Verlet skins (NM_SKIN for LJ, NM_SKIN_AL for the EAM)
CFG=${CFG:-C4}; VAR=${VAR:-NM_SKIN_AL}
for rep in 1 2; do
for s in "$@"; do
  export $VAR=$s
  python bench.py --config $CFG --no-cpu --steps ${STEPS:-10} --warmup ${WARM:-8} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$CFG $VAR=$s  window %8.0f (%.3f ms)  sustained %8.0f (%.3f ms) rebuilds/sweep %.2f' % (d['window']['value'], d['window']['kernel_avg_ms'], d['sustained']['value'], d['sustained']['kernel_avg_ms'], d['sustained']['list_rebuilds_per_sweep']))"
done
done
