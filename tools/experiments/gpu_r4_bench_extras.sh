#!/bin/bash
# round 4, pass g: the bench line with all extras; wall time of the whole run
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r4g}
mkdir -p $O
cd $R
T0=$(date +%s.%N); python bench.py > $O/bench.json 2> $O/bench.err; T1=$(date +%s.%N); echo "bench wall $(echo "$T1 - $T0" | bc) s"
tail -3 $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value %.0f ms/step %.3f dtype %s" % (d["value"], d["ms_per_step"], d["dtype"]))
print(d["config"]["arithmetic"][:100])
ex=d.get("extra",{})
for k in ("voxelize","train_step","n500_f320_nf3"):
    print(k, json.dumps(ex.get(k))[:1800])
print("cpu", json.dumps(d.get("cpu_baseline"))[:300])
print({k:(v.get("value") if isinstance(v,dict) else v) for k,v in ex.get("car_90_320_3",{}).items() if k!="config"})
PY
