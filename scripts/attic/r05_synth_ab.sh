#!/bin/bash
# synthetic 100k x 500k (BASELINE.json configs[4]), same box: the class sort inside windows of the record stream (TB_GLOBAL_SORT_WINDOW), hot tier and workgroup teams
cd $GRAFT_REPO_ROOT
common="--workload synthetic --steps 2 --warmup 1 --side-steps 0 --other-steps 0 --reference-seconds 0 --no-cpu-baseline"
run() { env "$@" timeout 300 python3 bench.py $common --fixpoint $FP 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$* $FP: nodes/s %.4e  propagations/s %.4e  evaluations/node %.0f' % (d['nodes_per_sec'], d['value'], d['value'] / d['nodes_per_sec']))
"; }
for FP in wac1 event ac1; do
  for W in 0 1024 4096 32768 1000000; do run TB_GLOBAL_SORT_WINDOW=$W; done
done
FP=wac1
for W in 0 4096; do
  run TB_GLOBAL_SORT_WINDOW=$W TB_TEAM=1 TB_TEAM_SPLIT=1 TB_TEAM_RELAXED=1
  run TB_GLOBAL_SORT_WINDOW=$W TB_TEAM=1 TB_TEAM_SPLIT=4 TB_TEAM_RELAXED=1
done
