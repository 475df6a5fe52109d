#!/bin/bash
# round 4, pass e: ablation builds of the fp16 K0 kernel (pure kernel time from rocprofv3 stats, B = 8)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r4e}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lib in $R/shasta_amd/csrc/libshasta_hip.so $R/tools/probes/_bin/libshasta_${2:-abl}_*.so; do
  n=$(basename $lib .so)
  SHASTA_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -o d -- python3 $R/tools/conv_only.py --batch 8 --iters 8 > $O/$n.log 2>&1
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/$n/d_kernel_stats.csv")))
for r in rows:
    if "conv_f16" in r["Name"]: print("%-28s %-40s calls %3s avg_us %9.1f min %9.1f" % ("$n", r["Name"][:40], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
done
