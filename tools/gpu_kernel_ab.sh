#!/bin/bash
# A/B of one kernel across library variants on ONE box: mean / min launch time of the kernels whose name contains <substring> (full-batch
# launches only) in bench runs at 1024 frame-pairs per step, two alternating rounds.
# usage (GPU box): bash tools/gpu_kernel_ab.sh <subdir of gpurun_out> <kernel substring> <min grid size> <variant lib> ...
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
K=$2
G=$3
shift 3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for lib in "" "$@"; do
  n=default; [ -n "$lib" ] && n=$(basename $lib .so)
  rm -rf $O/ab_$n
  SHASTA_HIP_LIB=${lib:+$R/$lib} rocprofv3 --kernel-trace --output-format csv -d $O/ab_$n -o p -- python3 $R/bench.py --batch 1024 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/ab_$n.log 2>&1
  python3 - <<PY
import csv,glob,json
for f in glob.glob("$O/ab_$n/**/*kernel_trace.csv", recursive=True):
    d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6 for r in csv.DictReader(open(f)) if "$K" in r["Kernel_Name"] and int(r["Grid_Size_X"]) >= $G]
    if d: print("$n: $K mean %.4f ms min %.4f (n=%d)" % (sum(d)/len(d), min(d), len(d)))
lines=[l for l in open("$O/ab_$n.log") if l.startswith("{")]
if lines: print("   value", round(json.loads(lines[-1])["value"] or json.loads(lines[-1]).get("probe_value", 0)))
PY
done
done
