#!/bin/bash
# A/B of the one-pass aff kernel's hardening (tickets, release / acquire fences) on one box: kernel time of aff_frame16_kernel in a
# bench run at 1024 frame-pairs per step, for the shipped library and variants built by tools/build_variant.py.
# usage (GPU box): bash tools/gpu_aff_ab.sh <subdir of gpurun_out> <variant lib> ...
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for lib in "" "$@"; do
  n=default; [ -n "$lib" ] && n=$(basename $lib .so)
  rm -rf $O/ab_$n
  SHASTA_HIP_LIB=${lib:+$R/$lib} rocprofv3 --kernel-trace --output-format csv -d $O/ab_$n -o p -- python3 $R/bench.py --batch 1024 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/ab_$n.log 2>&1
  python3 - <<PY
import csv,glob
for f in glob.glob("$O/ab_$n/**/*kernel_trace.csv", recursive=True):
    d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6 for r in csv.DictReader(open(f)) if "aff_frame16" in r["Kernel_Name"] and int(r["Grid_Size_X"]) > 100000]
    if d: print("$n: aff_frame16 mean %.4f ms min %.4f (n=%d)" % (sum(d)/len(d), min(d), len(d)))
PY
done
done
