#!/bin/bash
# kernel profile of the shipped car configuration (max_obj 90, F = 320, nf = 3) at 512, 64, 8 and 1 frame-pairs per step
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3car
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for b in 512 8 1; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b$b -o d -- python3 $R/tools/time_forward.py --batch $b --steps 40 > $O/prof_b$b.log 2>&1
grep -h "ms" $O/prof_b$b.log | tail -2
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/prof_b$b/d_kernel_stats.csv")))
print("B=$b")
for r in rows[:18]:
    print("%-66s calls %4s avg_us %9.1f pct %5s" % (r["Name"][:66], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
done
