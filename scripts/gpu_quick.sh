#!/bin/bash
# quick A/B on the GPU box: headline trees + parity subset, then the headline bench without side rows
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=${1:-q}
timeout 600 python3 -m pytest tests/test_headline_trees.py tests/test_gpu_parity.py tests/test_gpu_selfcheck.py -x -q > gpurun_out/${tag}_t.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${tag}_t.log
timeout 300 python3 bench.py --steps 3 --warmup 1 --side-steps 0 --no-cpu-baseline --reference-seconds 0 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
python3 - <<PY
import json
d=json.load(open("gpurun_out/${tag}_bench.json"))
print("nodes/s %.4e  props/s %.4e  ms/step %.1f" % (d["nodes_per_sec"], d["value"], d["ms_per_step"]))
PY
