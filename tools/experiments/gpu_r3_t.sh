#!/bin/bash
cd "$(dirname "$0")/../.." && R=$PWD
for b in ${BATCHES:-256 128}; do
echo "== B=$b"
SHASTA_HIP_LIB=$R/tools/probes/_bin/libshasta_pstamp.so timeout 600 python3 tools/pair_clock.py $b 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|waves of the first"
done
