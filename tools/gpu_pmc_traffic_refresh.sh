#!/bin/bash
# Re-collects the FETCH_SIZE / WRITE_SIZE passes behind profiles/pmc_traffic.json into an existing measurement directory.
# usage (GPU box): bash tools/gpu_pmc_traffic_refresh.sh <subdir of gpurun_out>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-final}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for b in 1 128 512 1024; do
  rm -rf $O/pmc_fetch_b$b $O/pmc_write_b$b
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_b$b -o p -- python3 $R/bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc_fetch_b$b.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_b$b -o p -- python3 $R/bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc_write_b$b.log 2>&1
done
rm -rf $O/prof_b1024 $O/prof_default
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o d -- python3 $R/bench.py --batch 512 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/prof_default.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b1024 -o d -- python3 $R/bench.py --batch 1024 --steps 12 --warmup 3 --no-cpu-baseline --no-extras > $O/prof_b1024.log 2>&1
ls $O | head -3
