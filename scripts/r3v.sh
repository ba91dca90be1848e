#!/bin/bash
O=gpurun_out/r3v; mkdir -p $O
for v in default ab9; do
  if [ $v = default ]; then unset NM_HIP_LIB; else export NM_HIP_LIB=$PWD/neuralmelting_amd/libnm_hip_ab9.so; fi
  timeout -k 10 400 python bench.py --config C5 --rows 2 --warmup 8 --steps 6 --no-cpu > $O/bench_$v.json 2> $O/bench_$v.err; python -c "
import json; d=json.load(open('$O/bench_$v.json')); print('$v: Q=%d window %.0f (%.1f ms) sustained %.0f (%.1f ms) slot mean/max %.1f/%.1f' % (d['roofline']['cus_per_replica'], d['window']['value'], d['window']['kernel_avg_ms'], d['sustained']['value'], d['sustained']['kernel_avg_ms'], d['sustained']['slot_block_ms_mean'], d['sustained']['slot_block_ms_max']))"; tail -1 $O/bench_$v.err
done
NM_HIP_LIB=$PWD/neuralmelting_amd/libnm_hip_ab9.so NM_CUS_PER_REPLICA=4 timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "large_cells and 8-2" 2>&1 | tail -2
