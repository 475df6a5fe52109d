#!/bin/bash
# round 3: NaN propagation tests + A/B of the NaN-propagating ReLU build against the previous commit's library
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3g
mkdir -p $O
cd $R
python -m pytest tests/test_hip_parity.py -m gpu -q --tb=short -rf -x -k "non_finite or golden or pair_kernels or piece_kernels or operating" > $O/pytest.log 2>&1
grep -n "passed\|failed" $O/pytest.log | tail -3; grep -n "^FAILED\|^E " $O/pytest.log | head -20
AB_ROUNDS=2 bash tools/gpu_ab.sh r3g_ab tools/probes/_bin/libshasta_prev.so shasta_amd/csrc/libshasta_hip.so
