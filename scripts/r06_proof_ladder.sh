#!/bin/bash
# how long the proof of `objective <= B` takes on one MI355X for larger B (calibration of the sharded_search record: a longer fixed-work run scales more cleanly to 8 GPUs)
cd $GRAFT_REPO_ROOT
for B in 600 700 800 900; do
  timeout 120 python3 bench.py --mode solve --fixed-bound $B --solve-timeout 30 --no-cpu-baseline > /tmp/p.json 2>/tmp/p.err
  python3 -c "
import json; d=json.load(open('/tmp/p.json')); p=d.get('proof',{}); print('B=$B', {k:p.get(k) for k in ('seconds','exhaustive','nodes','eps_solved','eps_skipped','every_subproblem_accounted_once')})" 2>&1 | tail -1
done
