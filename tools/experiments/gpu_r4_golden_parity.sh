#!/bin/bash
# round 4, pass h: parity tests on the reference goldens (incl. the moderately sharp and heavy-tailed ones) + pointer cache / companion checks
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r4h}
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/test_hip_parity.py -m gpu -q -x --tb=short -k "${2:-golden or headline_size or pointer_cache or companion or zz_argmax}" -rP > $O/pytest.log 2>&1
grep -E "passed|failed|arg-max agreement|max\|m1-ref|Error|assert" $O/pytest.log | cut -c1-400 | tail -40
