#!/bin/bash
# round 4: the configs 2-4 chain - parity tests of the frame-major path, then wall time per stage for each loader mode
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r4i}
mkdir -p $O
cd $R
echo skip tests

for mode in none thread; do
  timeout 90 python tools/time_pipeline.py --scenes 20 --frames 40 --batch 40 --sync 0 --prefetch $mode > $O/pipe_$mode.log 2>&1
  echo "== prefetch $mode"; tail -2 $O/pipe_$mode.log | cut -c1-330
done
timeout 90 python tools/time_pipeline.py --scenes 20 --frames 40 --batch 40 --sync 1 --prefetch none > $O/pipe_sync.log 2>&1
tail -1 $O/pipe_sync.log | cut -c1-500
