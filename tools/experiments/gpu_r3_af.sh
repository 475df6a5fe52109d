#!/bin/bash
cd "$(dirname "$0")/../.." && R=$PWD
python -m pytest tests/test_hip_parity.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -6
python3 tools/car_b1.py 1 300 2>&1 | grep "car B"
python3 tools/car_b1.py 8 300 2>&1 | grep "car B"
python3 bench.py --batch 1 --steps 300 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=500 b1: %.4f ms' % d['ms_per_step'])"
