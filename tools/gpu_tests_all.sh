#!/bin/bash
# the whole GPU suite + smoke, as the driver runs them at round end
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-tests_all}
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q --tb=short -rf > $O/pytest.log 2>&1
grep -n "passed\|failed\|arg-max agreement" $O/pytest.log | tail -5
grep -n "^FAILED\|^ERROR" $O/pytest.log | head -20
python __graft_entry__.py smoke 2>&1 | tail -1
