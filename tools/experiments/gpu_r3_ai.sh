#!/bin/bash
cd "$(dirname "$0")/../.." && R=$PWD
O=$R/gpurun_out/r3ai; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for b in 64 512; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b$b -o d -- python3 $R/bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/prof_b$b.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/prof_b$b/*kernel_stats.csv")[0]
print("B=$b")
for r in csv.DictReader(open(f)):
    if int(r["Calls"])>=20 and "shasta" in r["Name"] and float(r["AverageNs"])<150000:
        print("  %-70s calls %4d avg %8.2f us" % (r["Name"][:70], int(r["Calls"]), float(r["AverageNs"])/1e3))
PY
done
