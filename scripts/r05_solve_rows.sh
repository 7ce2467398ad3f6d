#!/bin/bash
# bench.py --mode solve at 1 rank and at 2 ranks sharing the one GPU of the box, for profiles/r05_solve_mode.json
cd $GRAFT_REPO_ROOT
out=gpurun_out/r05_solve_mode.jsonl; : > $out
for w in wordpress7_500 trains15 accap_a3; do
  timeout 300 python3 bench.py --mode solve --workload $w --solve-timeout 60 2>/dev/null | grep '^{' >> $out
  timeout 400 python3 bench.py --gpus 2 --share-device --dist-backend gloo --mode solve --workload $w --solve-timeout 60 2>/dev/null | grep '^{' >> $out
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r05_solve_mode.jsonl"):
    d = json.loads(l)
    for k in ("proof", "to_target"):
        r = d.get(k)
        if r: print(d["workload"], "ranks", d["n_gpus"], k, {x: r.get(x) for x in ("seconds", "time_to_target_s", "exhaustive", "has_solution", "best_objective_bound", "nodes", "eps_solved", "eps_skipped", "stolen_subproblems", "every_subproblem_accounted_once", "linked")})
PY
