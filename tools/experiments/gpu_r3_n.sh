#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3n
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -q --tb=short -x -s -k "fixed_grid or pair_kernels or fused_row or operating" > $O/pytest.log 2>&1
grep -n "passed\|failed\|f16grid\|Error\|error" $O/pytest.log | cut -c1-400 | tail -20
for a in f16x2 f16grid; do
python bench.py --no-cpu-baseline --no-extras --steps 30 --arithmetic $a > $O/bench_$a.json 2> $O/bench_$a.err
python - <<PY
import json
d=json.loads(open("$O/bench_$a.json").read().strip().splitlines()[-1])
r=[d["roofline"],d["roofline_second"]]
p=[x for x in r if x["kernel"].startswith("pair")][0]; l=[x for x in r if not x["kernel"].startswith("pair")][0]
e=d.get("energy") or {"joules_per_step": float("nan"), "avg_power_w": float("nan")}
print("%-10s %.0f fp/s  step %.3f ms  pair %.3f ms  L1 %.3f ms  %.2f J/step  %.0f W  selfcheck %.1e" % ("$a", d["value"], d["ms_per_step"], p["avg_launch_ms"], l["avg_launch_ms"], e["joules_per_step"], e["avg_power_w"], d["selfcheck_max_abs"]))
PY
done
tail -3 $O/bench_f16grid.err
