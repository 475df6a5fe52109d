#!/bin/bash
# quick GPU pass while iterating on kernels: parity tests that touch the forward, a bench line, kernel stats
# usage: bash tools/gpu_quick.sh <subdir of gpurun_out> [pytest -k expression]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-quick}
mkdir -p $O
cd $R
python -m pytest tests/test_hip_parity.py tests/test_fuzz.py -m gpu -q -x --tb=short -k "${2:-forward or headline or batched or edge or operating or f32_arith or fuzz or large}" > $O/pytest.log 2>&1
tail -4 $O/pytest.log
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value %.0f ms/step %.3f selfcheck %.2e" % (d["value"], d["ms_per_step"], d["selfcheck_max_abs"]))
for k in ("roofline","roofline_second"):
    print(k, d[k]["kernel"][:30], "ms %.3f frac %.3f" % (d[k]["avg_launch_ms"], d[k]["frac"]))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o d -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/prof.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/prof/d_kernel_stats.csv")))
for r in rows[:16]:
    print("%-60s calls %4s avg_us %9.1f pct %5s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
