#!/bin/bash
# A/B of library variants on ONE box: bash tools/gpu_ab.sh <subdir> <lib1> <lib2> ...   (paths relative to the repo root)
# alternates the variants (A B A B) so that box / thermal drift shows up as spread, not as a difference
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
shift
mkdir -p $O
cd $R
for round in $(seq 1 ${AB_ROUNDS:-2}); do
  for lib in "$@"; do
    name=$(basename $lib .so)
    SHASTA_HIP_LIB=$R/$lib python bench.py --no-cpu-baseline --no-extras --steps ${AB_STEPS:-40} > $O/${name}_$round.json 2> $O/${name}_$round.err
    python - <<PY
import json
d=json.loads(open("$O/${name}_$round.json").read().strip().splitlines()[-1])
r=[d["roofline"],d["roofline_second"]]
p=[x for x in r if x["kernel"].startswith("pair")][0]; l=[x for x in r if not x["kernel"].startswith("pair")][0]
e=d.get("energy") or {"joules_per_step": float("nan"), "avg_power_w": float("nan")}
print("%-28s round $round: %.0f fp/s  step %.3f ms  pair %.3f ms  L1 %.3f ms  %.2f J/step  %.0f W  selfcheck %.1e" % ("$name", d["value"], d["ms_per_step"], p["avg_launch_ms"], l["avg_launch_ms"], e["joules_per_step"], e["avg_power_w"], d["selfcheck_max_abs"]))
PY
  done
done
