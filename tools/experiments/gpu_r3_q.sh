#!/bin/bash
# full bench line at the new default batch + the two operating-point tests
cd "$(dirname "$0")/../.." && R=$PWD
O=$R/gpurun_out/r3q; mkdir -p $O

s=$(date +%s); python3 bench.py > $O/bench_full.json 2> $O/bench_full.err
echo "bench wall $(( $(date +%s) - s )) s"
python3 - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r3q/bench_full.json") if l.startswith("{")][-1])
print({k:d[k] for k in ("value","ms_per_step","config","selfcheck_max_abs") if k in d})
print(d["roofline"]); print(d.get("cpu_baseline"))
for k,v in d["extra"].items():
    print(k, json.dumps(v)[:600])
PY
