#!/bin/bash
# long runs of the bench presets: thousands of launches, every one checked for status bits at the closing synchronisation
O=gpurun_out/soak; mkdir -p $O
timeout -k 10 300 python bench.py --steps 1500 --warmup 5 --no-cpu > $O/C2.json 2> $O/C2.err; echo "C2 rc=$?"
timeout -k 10 300 python bench.py --config C4 --steps 400 --no-cpu > $O/C4.json 2> $O/C4.err; echo "C4 rc=$?"
timeout -k 10 300 python bench.py --config C3 --steps 300 --no-cpu > $O/C3.json 2> $O/C3.err; echo "C3 rc=$?"
NM_OVERSUBSCRIBE=1 timeout -k 10 400 python bench.py --config C5 --steps 60 --no-cpu > $O/C5x2.json 2> $O/C5x2.err; echo "C5x2 rc=$?"
python - <<'PY'
import json
for f in ("C2", "C4", "C3", "C5x2"):
    d = json.load(open("gpurun_out/soak/%s.json" % f))
    print(f, d["steps"], "timed steps per region: sustained %.0f window %.0f sweeps/s, Q = %d, equilibration cycles %s" % (d["value"], d["window"]["value"], d["roofline"]["cus_per_replica"], d.get("equilibration_cycles")))
PY
