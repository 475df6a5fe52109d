#!/bin/bash
cd "$(dirname "$0")/../.." && R=$PWD
s=$(date +%s); python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/driver_like.json 2> gpurun_out/driver_like.err; echo "rc=$? wall $(( $(date +%s) - s )) s"
python3 - <<'PY'
import json
lines=[l for l in open("gpurun_out/driver_like.json") if l.strip()]
print(len(lines), "line(s) on stdout")
d=json.loads(lines[-1])
print({k:d[k] for k in ("metric","value","unit","n_gpus","steps","warmup","ms_per_step","higher_is_better","scaling","vs_baseline","dtype","data")})
print(d["roofline"]["bound"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_source"][:60])
print(d["cpu_baseline"])
print(list(d["extra"].keys()))
PY
tail -3 gpurun_out/driver_like.err
