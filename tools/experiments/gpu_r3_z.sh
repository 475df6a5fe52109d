#!/bin/bash
cd "$(dirname "$0")/../.." && R=$PWD
for b in 512 1024; do
SHASTA_HIP_LIB=$R/tools/probes/_bin/libshasta_l1tl.so timeout 600 python3 tools/l1_timeline.py $b 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"
done
