#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash scripts/gpu_ab3.sh ab_ref.so libturbo_hip.so ab_ref.so libturbo_hip.so
timeout 1800 python3 -m pytest tests -m gpu -x -q > gpurun_out/q_t.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/q_t.log
