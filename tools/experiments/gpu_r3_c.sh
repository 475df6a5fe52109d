#!/bin/bash
# round 3, third GPU pass: fused from-bev entry, two-column softmax, pre-cut weight stream
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c
mkdir -p $O
cd $R
python -m pytest tests/test_hip_parity.py tests/test_fuzz.py -m gpu -q --tb=short -rf -x > $O/pytest.log 2>&1
tail -8 $O/pytest.log
for rnd in 1 2; do
  for v in "" "--no-precut"; do
    python bench.py --no-cpu-baseline --no-extras --steps 40 $v > $O/bench_${rnd}_${v#--}.json 2> $O/bench_${rnd}_${v#--}.err
    python - <<PY
import json
d=json.loads(open("$O/bench_${rnd}_${v#--}.json").read().strip().splitlines()[-1])
r=[d["roofline"],d["roofline_second"]]
p=[x for x in r if x["kernel"].startswith("pair")][0]; l=[x for x in r if not x["kernel"].startswith("pair")][0]
e=d.get("energy") or {"joules_per_step": float("nan"), "avg_power_w": float("nan")}
print("%-10s round $rnd: %.0f fp/s  step %.3f ms  pair %.3f ms  L1 %.3f ms  %.2f J/step  %.0f W  selfcheck %.1e" % ("${v:-plain}", d["value"], d["ms_per_step"], p["avg_launch_ms"], l["avg_launch_ms"], e["joules_per_step"], e["avg_power_w"], d["selfcheck_max_abs"]))
PY
  done
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o d -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/prof.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/prof/d_kernel_stats.csv")))
for r in rows[:18]:
    print("%-60s calls %4s avg_us %9.1f pct %5s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
