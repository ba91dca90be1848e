#!/bin/bash
# soak: every preset far into its equilibrated regime, the driver with outputs on, and the pipeline stages behind it
O=gpurun_out/r3u; mkdir -p $O
for c in "C2 120 20" "C4 100 10" "C3 80 8" "C5 60 4" "runsh 40 3"; do set -- $c; timeout -k 10 500 python scripts/probe_balance.py $1 $2 $3 > $O/soak_$1.txt 2>&1; echo "soak $1 rc=$?"; head -1 $O/soak_$1.txt; done
mkdir -p $O/run && cd $O/run && timeout -k 10 300 python -m neuralmelting_amd.remcmc -v -bm -n soak -e LJ -ss 4 -pn 8 -tn 8 -sn 96 -sm 128 -sc 64 -rd 32 > ../driver.log 2>&1; echo "driver rc=$?"; tail -2 ../driver.log; ls -la | head; cd - > /dev/null
