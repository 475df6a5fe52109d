#!/bin/bash
cd "$(dirname "$0")/../.." && R=$PWD
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -6
python3 __graft_entry__.py smoke 2>&1 | tail -1
python3 tools/car_b1.py 1 300 2>&1 | grep "car B"
python3 tools/car_b1.py 8 300 2>&1 | grep "car B"
