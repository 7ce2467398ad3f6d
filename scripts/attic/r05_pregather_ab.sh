#!/bin/bash
# Operands gathered one slice ahead in the sweeps over global-memory stores (kernels.hpp: PREG), and two team workgroups per CU: same-box A/B of library variants
#   scripts/r05_pregather_ab.sh "<lib> <lib> ..."   (paths under turbo_amd/lib)
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $root
for lib in $1; do
  echo "== $lib"
  TURBO_HIP_LIB=$root/turbo_amd/lib/$lib TEAM_NO_PMC=1 TEAM_FPS="${AB_FPS:-wac1 ac1}" TEAM_CFGS="${AB_CFGS:-1:4:1 0:1:0}" bash scripts/r05_team_ab.sh
done
