#!/bin/bash
# round 4, pass i: the configs 2-4 chain - parity tests of the frame-major path, then wall time per stage
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r4i}
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_pipeline.py tests/test_decode_golden.py tests/test_pub_tracker.py -m gpu -q -x --tb=short > $O/pytest.log 2>&1
tail -4 $O/pytest.log
timeout 600 python tools/time_pipeline.py --scenes 20 --frames 40 --batch 40 > $O/pipe_sync.log 2>&1
tail -3 $O/pipe_sync.log
timeout 600 python tools/time_pipeline.py --scenes 20 --frames 40 --batch 40 --sync 0 > $O/pipe_nosync.log 2>&1
tail -2 $O/pipe_nosync.log
