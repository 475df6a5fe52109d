#!/bin/bash
# round 3 final measurement pass: the whole GPU suite once more, then tools/measure_round.sh (bench lines, kernel stats, PMC passes)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r3m2}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q --tb=short -rf > $O/pytest.log 2>&1
grep -n "passed\|failed" $O/pytest.log | tail -3
python __graft_entry__.py smoke 2>&1 | tail -1
bash tools/measure_round.sh $TAG 2>&1 | tail -14
