#!/bin/bash
O=gpurun_out/r3b; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
NM_HIP_LIB=$PWD/neuralmelting_amd/libnm_hip_prof.so timeout -k 10 300 python scripts/check_bounds.py > $O/bounds.txt 2>&1; echo "bounds rc=$?"; tail -2 $O/bounds.txt
timeout -k 10 300 python bench.py --warmup 5 --steps 20 > $O/bench_C2.json 2> $O/bench_C2.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --warmup 5 --steps 10 --iterative --cpu-seconds 4 > $O/bench_C2_iter.json 2> $O/bench_C2_iter.err; echo "bench iter rc=$?"
timeout -k 10 200 python scripts/probe_balance.py C3 30 6 > $O/balance_C3_eq.txt 2>&1
timeout -k 10 300 python scripts/probe_balance.py runsh 20 3 > $O/balance_runsh_eq.txt 2>&1
