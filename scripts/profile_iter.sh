#!/bin/bash
# kernel statistics of the reference's default (iterative) position moves on the C2 grid, cycles 3-7 (the only regime that mode has), run on the GPU box:
#   scripts/profile_iter.sh -> gpurun_out/prof_C2_iter/r03_kernel_stats_C2_iter.csv + the section shares of the diagnostic build
ROOT=$PWD; OUT=$ROOT/gpurun_out/prof_C2_iter; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2 -- python3 $ROOT/bench.py --iterative --no-cpu > $OUT/bench_iter.json 2> $OUT/stats2.err
cd $ROOT
cp $(find $OUT/stats2 -name '*kernel_stats.csv' | head -1) $OUT/r03_kernel_stats_C2_iter.csv
head -4 $OUT/r03_kernel_stats_C2_iter.csv
NM_PROBE_ITER=1 timeout -k 10 200 python scripts/probe_sections.py 4 8 8 128 5 3 > $OUT/r03_sections_C2_iter.txt 2>&1; cat $OUT/r03_sections_C2_iter.txt
python -c "
import json; d=json.load(open('$OUT/bench_iter.json')); print('bench --iterative under the profiler: %.0f sweeps/s, kernel %.2f ms' % (d['value'], d['roofline']['kernel_avg_ms']))"
