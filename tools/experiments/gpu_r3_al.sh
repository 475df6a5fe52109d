#!/bin/bash
# low pieces by v_fma_mixlo / mixhi_f16 (variant mixlo) against the default: accuracy test with the variant, then A/B
cd "$(dirname "$0")/../.." && R=$PWD
SHASTA_HIP_LIB=$R/tools/probes/_bin/libshasta_mixlo.so python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "pair or operating_points or golden or odd_table" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
AB_STEPS=30 AB_ROUNDS=2 bash tools/gpu_ab.sh r3al shasta_amd/csrc/libshasta_hip.so tools/probes/_bin/libshasta_mixlo.so
