#!/bin/bash
# Class sort inside windows of the record stream (TB_GLOBAL_SORT_WINDOW, engine.hip: to_internal) under workgroup teams: the team kernel is VALU bound (profiles/r05_team_pmc.json),
# and a mixed slice pays every class body -- same box, synthetic 100k x 500k
root=$GRAFT_REPO_ROOT
cd $root
for w in ${WINDOWS:-0 256 1024 4096 0 1024}; do
  echo "== window $w"
  TB_GLOBAL_SORT_WINDOW=$w TEAM_NO_PMC=1 TEAM_FPS="${AB_FPS:-wac1 ac1}" TEAM_CFGS="${AB_CFGS:-1:4:1}" bash scripts/r05_team_ab.sh
done
