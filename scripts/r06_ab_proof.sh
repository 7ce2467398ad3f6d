#!/bin/bash
# same-box A/B of engine builds on BOTH searches of the headline instance: the branch-and-bound step (scripts/r06_ab.sh) and the proof of the sharded_search record
cd $GRAFT_REPO_ROOT
AB_NO_PMC=${AB_NO_PMC:-1} bash scripts/r06_ab.sh "$@"
for pass in 1 2; do for lib in "$@"; do
  export TURBO_HIP_LIB=$GRAFT_REPO_ROOT/turbo_amd/lib/$lib
  timeout 200 python3 scripts/valu_by_phase.py 0x0 wordpress7_500 0 proof:500:21 2>&1 | tail -1 | sed "s#^#$lib pass $pass proof: #"
done; done
