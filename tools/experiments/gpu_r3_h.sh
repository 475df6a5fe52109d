#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}  # (run from the repo root)
cd $R
python -m pytest tests/test_hip_parity.py -m gpu -q --tb=short -x -k "non_finite or golden" 2>&1 | tail -2
AB_ROUNDS=2 bash tools/gpu_ab.sh r3h_ab tools/probes/_bin/libshasta_prev.so shasta_amd/csrc/libshasta_hip.so tools/probes/_bin/libshasta_pace0.so tools/probes/_bin/libshasta_pace8.so
