#!/bin/bash
# usage: scripts/pmc_quick.sh <outdir> -- <program args...>; two SQ passes + per-wave-propagation ratios
out=$1; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $out/p1 -o p -- "$@" > $out/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS -d $out/p2 -o p -- "$@" > $out/p2.log 2>&1
python3 scripts/pmc_summary.py $out
grep -o '"value": [0-9.e+]*\|"avg_launch_ms": [0-9.]*' $out/p1.log | head -2
