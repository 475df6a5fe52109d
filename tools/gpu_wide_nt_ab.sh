#!/bin/bash
# A/B of a library variant on the headline: alternating bench runs + FETCH_SIZE of the weight stream.  bash tools/gpu_wide_nt_ab.sh <variant lib>
R=$GRAFT_REPO_ROOT
V=$R/$1
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do
  python3 $R/bench.py --no-cpu-baseline --no-extras --steps 30 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', round(d['value']), round(d['roofline_second']['avg_launch_ms'],3))"
  SHASTA_HIP_LIB=$V SHASTA_BENCH_PROBE=1 python3 $R/bench.py --no-cpu-baseline --no-extras --steps 30 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant (probe: no assertions)', round(d['probe_value']), round(d['roofline_second']['avg_launch_ms'],3))"
done
for lib in "" $V; do
  rm -rf /tmp/pm; SHASTA_HIP_LIB=$lib SHASTA_BENCH_PROBE=${lib:+1} rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pm -o p -- python3 $R/bench.py --batch 1024 --steps 4 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
  python3 - <<PY
import csv,glob
v=[float(r["Counter_Value"]) for f in glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "anchor_l1_wide" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
print("${lib:-default} weight stream FETCH x2: %.2f GB" % (2*1024*sum(v)/len(v)/1e9))
PY
done
