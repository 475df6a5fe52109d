#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests/test_hip_parity.py tests/test_fuzz.py -m gpu -q --tb=short -x -k "golden or fuzz or edge or batched" 2>&1 | tail -2
python tools/time_forward.py --batch 1 8 64 512 --steps 200 2>&1 | grep ms
python tools/time_forward.py --max-obj 20 --batch 1 8 512 --steps 200 2>&1 | grep ms
python tools/time_forward.py --max-obj 500 --feats 7 --points 4 --batch 1 8 --steps 100 2>&1 | grep ms
