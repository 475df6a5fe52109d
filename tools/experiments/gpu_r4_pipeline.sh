#!/bin/bash
# round 4: configs 2-4 chain - parity tests, frames/s without and with the stage timer, host-side profile
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r4pipe}
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_pipeline.py tests/test_pub_tracker.py tests/test_frames.py -m gpu -q -x --tb=short > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 600 python tools/time_pipeline.py --sync 0 > $O/pipe.log 2>&1
timeout 600 python tools/time_pipeline.py --sync 1 > $O/pipe_sync.log 2>&1
grep -h frames_per_s $O/pipe.log $O/pipe_sync.log | cut -c1-120
timeout 600 python tools/profile_pipeline_host.py --top 25 > $O/host_profile.txt 2>&1
