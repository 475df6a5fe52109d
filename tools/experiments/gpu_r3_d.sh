#!/bin/bash
# round 3, fourth GPU pass: gather absmax (spread lines), pre-cut default, XCD-aware pair mapping A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3d
mkdir -p $O
cd $R
python -m pytest tests/test_hip_parity.py -m gpu -q --tb=short -rf -x -k "fused or precut or aux or golden or operating or headline" > $O/pytest.log 2>&1
tail -5 $O/pytest.log
AB_ROUNDS=2 bash tools/gpu_ab.sh r3d_ab shasta_amd/csrc/libshasta_hip.so tools/probes/_bin/libshasta_xcd.so
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o d -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/prof.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/prof/d_kernel_stats.csv")))
for r in rows[:14]:
    print("%-60s calls %4s avg_us %9.1f pct %5s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
