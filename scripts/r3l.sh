#!/bin/bash
O=gpurun_out/r3l; mkdir -p $O
timeout -k 10 300 python scripts/iter_watch.py 40 > $O/iter_new.txt 2>&1; tail -25 $O/iter_new.txt
(cd wt_old && timeout -k 10 300 python ../scripts/iter_watch.py 40 > ../$O/iter_old.txt 2>&1); tail -25 $O/iter_old.txt
