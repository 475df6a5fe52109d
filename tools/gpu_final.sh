#!/bin/bash
# final measurement pass of a round: the whole GPU suite + smoke, then tools/measure_round.sh (bench lines, kernel stats, PMC passes)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r4m1}
bash $R/tools/gpu_tests_all.sh $TAG
bash $R/tools/measure_round.sh $TAG 2>&1 | tail -24
