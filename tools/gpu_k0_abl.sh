#!/bin/bash
# K0 (512-pixel tiles) ablations: kernel time + matrix-pipe counters per library variant.  bash tools/gpu_k0_abl.sh <subdir> <lib> ...
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  n=$(basename $lib .so)
  export SHASTA_HIP_LIB=$R/$lib
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/$n -o p -- python3 $R/tools/conv_only.py --batch 8 --iters 6 --heads ${K0_HEADS:-1} > $O/$n.log 2>&1
  echo "== $n"
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list); dur=[]
for f in glob.glob("$O/$n/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "shared_conv_f16" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$O/$n/**/*kernel_trace.csv", recursive=True):
    dur=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6 for r in csv.DictReader(open(f)) if "shared_conv_f16" in r["Kernel_Name"]]
m={k:sum(v)/len(v) for k,v in acc.items()}
if dur and m:
    d=sum(dur)/len(dur); cyc=m["GRBM_GUI_ACTIVE"]/8
    print("kernel %.3f ms  clock %.2f GHz  matrix pipe busy %.1f %%  VALU insts/wave-cycle %.3f  wait %.1f %%" % (d, cyc/d/1e6, 100*m["SQ_VALU_MFMA_BUSY_CYCLES"]/1024/cyc, m["SQ_INSTS_VALU"]/m["SQ_WAVE_CYCLES"], 100*m["SQ_WAIT_INST_ANY"]/m["SQ_WAVE_CYCLES"]))
PY
done
