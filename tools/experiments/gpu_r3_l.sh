#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
AB_ROUNDS=1 AB_STEPS=20 bash tools/gpu_ab.sh r3l_ab shasta_amd/csrc/libshasta_hip.so tools/probes/_bin/libshasta_wpb4.so tools/probes/_bin/libshasta_wpb6.so
