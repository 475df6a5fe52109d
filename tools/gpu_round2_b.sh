#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2b
mkdir -p $O
cd $R
python -m pytest tests/test_hip_parity.py tests/test_train_track.py tests/test_training.py -m gpu -q -rA --tb=short -s 2>&1 > $O/pytest_full.log
grep -E "^(PASSED|FAILED|ERROR)|passed|failed" $O/pytest_full.log | tail -80
