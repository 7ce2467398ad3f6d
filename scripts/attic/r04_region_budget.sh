#!/bin/bash
# GPU side of the instruction budget by region (scripts/region_budget.py): marker passes of the tuning build, SQ_INSTS_* per node of the production build
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TURBO_HIP_LIB=turbo_amd/lib/libturbo_hip_tuning.so python3 scripts/region_budget.py census 12000000 > gpurun_out/r04_region_census.json 2> gpurun_out/r04_region_census.err; echo "census rc=$?"
mkdir -p gpurun_out/pmcab_r04
bash scripts/pmc_ab.sh libturbo_hip.so 2>&1 | tail -2 | tee gpurun_out/r04_region_pmc.txt
