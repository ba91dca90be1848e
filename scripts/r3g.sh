#!/bin/bash
O=gpurun_out/r3g; mkdir -p $O
timeout -k 10 300 python scripts/r3g.py > $O/miss.txt 2>&1; cat $O/miss.txt
timeout -k 10 200 python scripts/probe_timeline.py > $O/timeline_C2.txt 2>&1; head -24 $O/timeline_C2.txt
timeout -k 10 200 python scripts/probe_sections.py 4 8 8 128 10 30 > $O/sections_C2_eq.txt 2>&1; cat $O/sections_C2_eq.txt
NM_PROBE_ITER=1 timeout -k 10 200 python scripts/probe_sections.py 4 8 8 128 8 3 > $O/sections_C2_iter_early.txt 2>&1; cat $O/sections_C2_iter_early.txt
NM_PROBE_ITER=1 timeout -k 10 200 python scripts/probe_sections.py 4 8 8 128 8 22 > $O/sections_C2_iter_late.txt 2>&1; cat $O/sections_C2_iter_late.txt
