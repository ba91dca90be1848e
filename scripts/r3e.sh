#!/bin/bash
O=gpurun_out/r3e; mkdir -p $O
./scripts/ubench_int16 > $O/ubench_int16.txt 2>&1; cat $O/ubench_int16.txt
NM_HIP_LIB=$PWD/neuralmelting_amd/libnm_hip_prof.so timeout -k 10 600 python scripts/check_bounds.py > $O/bounds.txt 2>&1; echo "bounds rc=$?"; tail -16 $O/bounds.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 300 python bench.py --warmup 5 --steps 10 --iterative --cpu-seconds 3 > $O/bench_C2_iter.json 2> $O/bench_C2_iter.err; echo "iter rc=$?"; tail -2 $O/bench_C2_iter.err
