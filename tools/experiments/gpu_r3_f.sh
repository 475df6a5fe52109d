#!/bin/bash
# round 3: 32-detection pair tiles (car / bus tables), precut threshold 17: whole GPU suite + the bench line with extras
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3f
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q --tb=short -rf -x > $O/pytest.log 2>&1
grep -n "passed\|failed" $O/pytest.log | tail -3
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -2 $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value %.0f ms/step %.3f selfcheck %.2e" % (d["value"], d["ms_per_step"], d["selfcheck_max_abs"]))
for k in ("roofline","roofline_second"):
    print(k, d[k]["kernel"][:50], "ms %.3f frac %.3f" % (d[k]["avg_launch_ms"], d[k]["frac"]))
def show(name, e):
    print("%-28s %9.0f fp/s %8.3f ms | %s %.3f ms frac %.3f | %s %.3f ms frac %.3f" % (name, e["value"], e["ms_per_step"], e["roofline"]["kernel"][:26], e["roofline"]["avg_launch_ms"], e["roofline"]["frac"], e["roofline_second"]["kernel"][:26], e["roofline_second"]["avg_launch_ms"], e["roofline_second"]["frac"]))
x=d["extra"]
for k,v in x["batch_sweep"].items(): show(k, v)
for k in ("arithmetic_f32","arithmetic_pieces"): show(k, x[k])
for k,v in x["car_90_320_3"].items():
    if isinstance(v, dict): show("car "+k, v)
PY
