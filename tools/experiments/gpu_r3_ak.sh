#!/bin/bash
cd "$(dirname "$0")/../.." && R=$PWD
python3 tools/warm_probe.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"
