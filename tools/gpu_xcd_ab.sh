#!/bin/bash
# A/B of the XCD-aware tile order of the pair kernel: time (tools/pair_time.py) and FETCH_SIZE with the default library and with
# tools/probes/_bin/libshasta_noxcd.so (python tools/build_variant.py noxcd pair_f16.hip -DPAIR_NO_XCD_ORDER).  usage (GPU box): bash tools/gpu_xcd_ab.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lib in "" tools/probes/_bin/libshasta_noxcd.so; do
  if [ -n "$lib" ]; then export SHASTA_HIP_LIB=$R/$lib; fi
  n=${lib:-default}
  python3 $R/tools/pair_time.py --max-obj 500 --points 4 --feats 7 --batch 512 --iters 20 --modes f16x2 2>&1 | tail -1 | cut -c1-120
  rm -rf /tmp/pm; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pm -o p -- python3 $R/tools/pair_time.py --max-obj 500 --points 4 --feats 7 --batch 512 --iters 3 --modes f16x2 > /dev/null 2>&1
  python3 - <<PY
import csv,glob
v=[float(r["Counter_Value"]) for f in glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "pair_f16_kernel" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
print("$n FETCH_SIZE per launch (KB, x2 on gfx950): %.0f -> %.2f GB" % (sum(v)/len(v), 2*1024*sum(v)/len(v)/1e9))
PY
done
