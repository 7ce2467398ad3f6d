#!/bin/bash
# usage: scripts/pmc_passes.sh <outdir> -- <program args...>   (run on the GPU box; counters only, no tracing domains)
out=$1; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH" \
           "SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $out/p$i -o p -- "$@" > $out/p$i.log 2>&1
done
