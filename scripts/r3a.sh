#!/bin/bash
# first GPU call of round 3: baseline tests, per-slot balance, section profile, both bench windows
O=gpurun_out/r3a; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" 
timeout -k 10 200 python scripts/probe_balance.py C2 30 10 > $O/balance_C2_eq.txt 2>&1 &&
timeout -k 10 200 python scripts/probe_balance.py C2 8 10 > $O/balance_C2_win.txt 2>&1 &&
NM_HIP_LIB=$PWD/neuralmelting_amd/libnm_hip_prof.so timeout -k 10 200 python scripts/probe_sections.py 4 8 8 128 10 30 > $O/sections_C2_eq.txt 2>&1 &&
timeout -k 10 200 python bench.py --warmup 30 --steps 10 --no-cpu > $O/bench_C2_eq.json 2> $O/bench_C2_eq.err &&
timeout -k 10 200 python bench.py --warmup 5 --steps 20 --no-cpu > $O/bench_C2_win.json 2> $O/bench_C2_win.err &&
timeout -k 10 300 python scripts/probe_balance.py C5 40 6 > $O/balance_C5_eq.txt 2>&1 &&
timeout -k 10 300 python scripts/probe_balance.py C4 30 10 > $O/balance_C4_eq.txt 2>&1
tail -3 $O/pytest.log; head -3 $O/balance_C2_eq.txt
