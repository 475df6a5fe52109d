#!/bin/bash
# round 4: configs 2-4 chain - parity tests, then frames/s with the frame files parsed in line / by 2, 3, 4 worker processes
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r4pipe}
mkdir -p $O
cd $R
nproc > $O/nproc.txt
timeout 900 python -m pytest tests/test_pipeline.py tests/test_pub_tracker.py tests/test_frames.py -m gpu -q -x --tb=short > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for w in 0 2 3 4; do
  timeout 600 python tools/time_pipeline.py --sync 0 --prefetch $w > $O/pipe_w$w.log 2>&1
  echo "== workers $w"; grep frames_per_s $O/pipe_w$w.log | cut -c1-120
done
timeout 600 python tools/time_pipeline.py --sync 1 --prefetch 3 > $O/pipe_sync_w3.log 2>&1
grep frames_per_s $O/pipe_sync_w3.log | tail -1
