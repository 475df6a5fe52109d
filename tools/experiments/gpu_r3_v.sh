#!/bin/bash
cd "$(dirname "$0")/../.." && R=$PWD
SHASTA_HIP_LIB=$R/tools/probes/_bin/libshasta_p4stamp.so timeout 600 python3 tools/pair_clock.py 512 pieces 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
