#!/bin/bash
cd $GRAFT_REPO_ROOT
export TURBO_HIP_LIB=$PWD/turbo_amd/lib/libturbo_hip_bounds.so
timeout 1500 python3 -m pytest tests/test_gpu_team.py tests/test_gpu_fullsize_global.py -m gpu -q -x -k "enumerate_every or default_plan or agree_with_single or full_size" 2>&1 | tail -4
unset TURBO_HIP_LIB
AB_NO_PMC=1 bash scripts/r06_ab.sh libturbo_hip.so ab/w6.so 2>&1 | tail -14
