#!/bin/bash
# round 3, first GPU pass: the whole GPU suite with the ABI-7 library, the bench line with its `extra` object, A/B of the pair variant
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3a
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q --tb=short -rf -x > $O/pytest.log 2>&1
tail -15 $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value %.0f ms/step %.3f selfcheck %.2e energy %s" % (d["value"], d["ms_per_step"], d["selfcheck_max_abs"], d["energy"]))
for k in ("roofline","roofline_second"):
    print(k, d[k]["kernel"][:40], "ms %.3f frac %.3f traffic %s" % (d[k]["avg_launch_ms"], d[k]["frac"], d[k]["traffic_source"]))
def show(name, e):
    print("%-28s %9.0f fp/s %8.3f ms | %s %.3f ms frac %.3f | %s %.3f ms frac %.3f" % (name, e["value"], e["ms_per_step"], e["roofline"]["kernel"][:22], e["roofline"]["avg_launch_ms"], e["roofline"]["frac"], e["roofline_second"]["kernel"][:22], e["roofline_second"]["avg_launch_ms"], e["roofline_second"]["frac"]))
x=d["extra"]
for k,v in x["batch_sweep"].items(): show(k, v)
for k in ("arithmetic_f32","arithmetic_pieces"): show(k, x[k])
for k,v in x["car_90_320_3"].items():
    if isinstance(v, dict): show("car "+k, v)
print("cpu", d.get("cpu_baseline"))
PY
AB_ROUNDS=2 bash tools/gpu_ab.sh r3a_ab shasta_amd/csrc/libshasta_hip.so tools/probes/_bin/libshasta_relu.so
