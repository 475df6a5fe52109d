#!/bin/bash
cd "$(dirname "$0")/../.." && R=$PWD
timeout 600 python3 tools/thermal_probe.py 512 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
