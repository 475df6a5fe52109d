#!/bin/bash
# matrix-pipe counters of the weight stream for library variants: bash tools/gpu_pmc_variant.sh <subdir> <lib> ...
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  n=$(basename $lib .so)
  SHASTA_HIP_LIB=$R/$lib SHASTA_BENCH_PROBE=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/$n -o p -- python3 $R/bench.py --batch 1024 --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $O/$n.log 2>&1
  echo "== $n"
  python3 $R/tools/pmc_table.py $O/$n anchor_l1_
  python3 - <<PY
import csv,glob
for f in glob.glob("$O/$n/**/*kernel_trace.csv", recursive=True):
    d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6 for r in csv.DictReader(open(f)) if "anchor_l1_wide" in r["Kernel_Name"] or "anchor_l1_split" in r["Kernel_Name"]]
    if d: print("duration ms: mean %.3f (n=%d)" % (sum(d)/len(d), len(d)))
PY
done
