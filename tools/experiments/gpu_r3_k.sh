#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -3
python tools/time_forward.py --batch 1 8 512 --steps 200 2>&1 | grep ms
python tools/time_forward.py --max-obj 500 --feats 7 --points 4 --batch 1 8 --steps 100 2>&1 | grep ms
python bench.py --no-cpu-baseline --no-extras --steps 30 | cut -c1-200
