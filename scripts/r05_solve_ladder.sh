#!/bin/bash
# Calibration of bench.py --mode solve on one GPU: which fixed bounds are refuted in seconds, how long a target takes.  usage: scripts/r05_solve_ladder.sh > gpurun_out/r05_solve_ladder.log
cd $GRAFT_REPO_ROOT
run() { timeout 120 python3 bench.py --mode solve --solve-timeout ${T:-15} "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        for k in ('proof', 'to_target'):
            if k in d:
                r = d[k]
                print('$*', k, {x: r.get(x) for x in ('seconds', 'time_to_target_s', 'exhaustive', 'has_solution', 'best_objective_bound', 'nodes', 'eps_solved', 'eps_skipped', 'subproblems_power', 'fixed_bound', 'target')})
"; }
for B in 500 1000 1500; do run --workload wordpress7_500 --fixed-bound $B --target 100000; done
for B in 0 10 20 30; do run --workload trains15 --fixed-bound $B --target 100000; done
for B in 20 40 55; do run --workload accap_a3 --subproblems-power 16 --fixed-bound $B --target 100000; done
T=30 run --workload wordpress7_500 --fixed-bound 500 --target 14000
T=30 run --workload trains15 --fixed-bound 0 --target 75
T=30 run --workload accap_a3 --subproblems-power 16 --fixed-bound 20 --target 135
