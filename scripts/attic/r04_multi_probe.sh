#!/bin/bash
# which two-rank bench configuration on one device completes subproblems AND steals inside a step (tests/test_gpu_multi.py)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for cfg in "trains15 10 64 2000000" "trains15 13 64 3000000" "trains15 16 256 3000000" "trains15 12 512 3000000" "accap_a3 14 128 3000000"; do
  set -- $cfg
  timeout 300 python3 bench.py --gpus 2 --share-device --dist-backend gloo --steps 1 --warmup 1 --workload $1 --subproblems-power $2 --or-nodes $3 --nodes-total $4 --no-cpu-baseline --side-steps 0 --other-steps 0 > gpurun_out/r04_multi_$1_$2_$3.json 2> gpurun_out/r04_multi_$1_$2_$3.err
  python3 - <<PY
import json
try:
    d=json.load(open("gpurun_out/r04_multi_$1_$2_$3.json")); m=d["multi_gpu"]
    print("$cfg: nodes/s %.3e solved %s skipped %s stolen %s skew_ms %s" % (d["nodes_per_sec"], m["eps_solved_per_step"], m["eps_skipped_per_step"], m["stolen_per_step"], m["start_skew_ms_max"]), [ (r["nodes"], r["kernel_ms"]) for r in m["per_rank"]])
except Exception as e:
    print("$cfg FAILED", e); print(open("gpurun_out/r04_multi_$1_$2_$3.err").read()[-800:])
PY
done
