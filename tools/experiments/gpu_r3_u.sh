#!/bin/bash
cd "$(dirname "$0")/../.." && R=$PWD
python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "pair or operating_points or batch or fixed_grid" 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
SHASTA_HIP_LIB=$R/tools/probes/_bin/libshasta_pstamp.so timeout 600 python3 tools/pair_clock.py 512 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
AB_STEPS=30 bash tools/gpu_ab.sh r3u shasta_amd/csrc/libshasta_hip.so tools/probes/_bin/libshasta_even.so tools/probes/_bin/libshasta_s55.so tools/probes/_bin/libshasta_s68.so
