#!/bin/bash
# round-3 refresh, part 2: A/B of the record prefetch in the two-wave configuration, then the instruction budget by phase
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash scripts/gpu_ab3.sh libturbo_hip.so ab_pf.so ab_pf6.so ab_w6.so libturbo_hip.so
timeout 900 python3 scripts/phase_budget.py r03 wordpress7_500 24000000 > gpurun_out/final2_phase.log 2>&1; echo "phase rc=$?"
tail -45 gpurun_out/final2_phase.log
