#!/bin/bash
cd "$(dirname "$0")/../.." && R=$PWD
AB_STEPS=30 AB_ROUNDS=2 bash tools/gpu_ab.sh r3x shasta_amd/csrc/libshasta_hip.so tools/probes/_bin/libshasta_prio_lo_hi.so tools/probes/_bin/libshasta_prio_hi_lo.so tools/probes/_bin/libshasta_prio_lo_hi_even.so tools/probes/_bin/libshasta_prio_lo_hi_s80.so tools/probes/_bin/libshasta_prio_hi_lo_even.so
