#!/bin/bash
# round 4, pass b: K0 fp16 form, where the two waves of a SIMD cut the next tile (variants built by tools/build_variant.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4b
mkdir -p $O
cd $R
python tools/conv_check.py --batches 1,8 > $O/base.jsonl 2> $O/base.err
grep -E '"time"|accuracy|equal' $O/base.jsonl | grep -v miopen | cut -c1-230
for lib in tools/probes/_bin/libshasta_c16_*.so; do
  echo "== $lib"
  SHASTA_HIP_LIB=$R/$lib python tools/conv_check.py --batches 1,8 > $O/$(basename $lib .so).jsonl 2>> $O/base.err
  grep -E '"time"' $O/$(basename $lib .so).jsonl | grep "f16x2" | cut -c1-200
  grep -E 'accuracy' $O/$(basename $lib .so).jsonl | head -1 | cut -c1-200
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o d -- python3 $R/tools/conv_check.py --batches 1,8 --iters 10 > $O/prof.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/prof/d_kernel_stats.csv")))
for r in rows[:10]:
    print("%-70s calls %4s avg_us %9.1f pct %5s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
