#!/bin/bash
# Root cause of r05's two-rank time-to-target rows ("what's weak" item 2): per-rank timelines of `bench.py --mode solve` with two ranks on ONE GPU --
# (a) as r05 ran it, a full grid per rank (the two persistent kernels cannot be co-resident), (b) as --share-device now plans it, half a grid per rank.
# Writes gpurun_out/r06_solve_timeline.jsonl (one bench line per run, tagged).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06_solve_timeline.jsonl; : > $out
run() {  # tag, then bench.py arguments
  tag=$1; shift
  line=$(timeout 200 python3 bench.py "$@" 2> gpurun_out/r06_tl_$tag.err | grep '^{' | tail -1)
  echo "{\"tag\": \"$tag\", \"rc\": $?, \"line\": ${line:-null}}" >> $out
  echo "$tag done"
}
for w in trains15 wordpress7_500; do
  run ${w}_1rank --mode solve --workload $w --solve-timeout 60
  run ${w}_2ranks_shared --gpus 2 --share-device --dist-backend gloo --mode solve --workload $w --solve-timeout 60
done
# r05's configuration: every rank launches the full grid (trains15: 3072 workgroups) on the one GPU
run trains15_2ranks_fullgrids --gpus 2 --share-device --dist-backend gloo --mode solve --workload trains15 --or-nodes 3072 --solve-timeout 40
python3 - <<'PY'
import json
for l in open("gpurun_out/r06_solve_timeline.jsonl"):
    r = json.loads(l)
    d = r["line"]
    if not d: print(r["tag"], "no line"); continue
    for k in ("proof", "to_target"):
        x = d.get(k)
        if not x: continue
        print(r["tag"], k, "seconds %.3f" % x["seconds"], "ttt", x["time_to_target_s"], "best", x["best_objective_bound"], "nodes %.3g" % x["nodes"], "once", x["every_subproblem_accounted_once"], "wg", x["workgroups_per_gpu"])
        for pr in x["per_rank"]:
            print("    rank %d kernel_ms %.1f nodes %.3g start_ret %.4f own_done %.3f loop_left %.3f best %s stolen %d solved %d" % (pr["rank"], pr["kernel_ms"], pr["nodes"], pr["t_start_returned_s"], pr["t_own_kernel_done_s"], pr["t_loop_left_s"], pr["best_bound"], pr["stolen_subproblems"], pr["eps_solved"]))
PY
