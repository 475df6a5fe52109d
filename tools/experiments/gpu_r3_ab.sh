#!/bin/bash
cd "$(dirname "$0")/../.." && R=$PWD
python -m pytest tests/test_hip_parity.py -m gpu -q -k "odd_table_sizes" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -30
