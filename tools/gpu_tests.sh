#!/bin/bash
# usage: bash tools/gpu_tests.sh <subdir> <pytest args...>
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
shift
mkdir -p $O
cd $R
python -m pytest "$@" -m gpu -q --tb=short -rf > $O/pytest.log 2>&1
tail -25 $O/pytest.log
