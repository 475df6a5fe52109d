#!/bin/bash
cd "$(dirname "$0")/../.." && R=$PWD
O=$R/gpurun_out/r3ae; mkdir -p $O


cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_car_b1 -o d -- python3 $R/tools/car_b1.py 1 300 > $O/prof_car_b1.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/prof_car_b1/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
tot=0
for r in rows:
    c=int(r["Calls"])
    if c>=300:
        per=float(r["TotalDurationNs"])/c*(c/ (620.0 if c>=600 else 310.0)) if False else float(r["AverageNs"])
        print("%-70s calls %5d avg %8.2f us" % (r["Name"][:70], c, float(r["AverageNs"])/1e3))
PY
