#!/bin/bash
# round 4, pass a: the fp16 form of K0 - accuracy against float64, timing against the f32 kernel and MIOpen, kernel stats
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4a
mkdir -p $O
cd $R
timeout 600 python tools/conv_check.py > $O/conv_check.jsonl 2> $O/conv_check.err
tail -30 $O/conv_check.jsonl; tail -5 $O/conv_check.err
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_fuzz.py -m gpu -q -x --tb=short -k "shared_conv" > $O/pytest_conv.log 2>&1
tail -5 $O/pytest_conv.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o d -- python3 $R/tools/conv_check.py --batches 1,8 --iters 10 > $O/prof.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/prof/d_kernel_stats.csv")))
for r in rows[:12]:
    print("%-70s calls %4s avg_us %9.1f pct %5s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
