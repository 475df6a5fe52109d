#!/bin/bash
cd "$(dirname "$0")/../.." && R=$PWD
python -m pytest tests/test_hip_parity.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -6
for round in 1 2; do
for lib in shasta_amd/csrc/libshasta_hip.so tools/probes/_bin/libshasta_nodirect.so; do
SHASTA_HIP_LIB=$R/$lib timeout 600 python3 tools/time_step.py 16 32 64 128 256 512 1024 2>&1 | grep "B="
done; done
