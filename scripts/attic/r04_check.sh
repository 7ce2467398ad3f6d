#!/bin/bash
# round 4 regression pass on the GPU box: the tests around the search kernels, then the headline bench (no side rows)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=${1:-r04chk}
timeout 1500 python3 -m pytest tests/test_gpu_fullgrid_paths.py tests/test_headline_trees.py tests/test_gpu_selfcheck.py tests/test_gpu_depth.py tests/test_gpu_constants.py tests/test_gpu_streaming.py tests/test_gpu_multi.py "tests/test_gpu_parity.py::test_element_models_tree_identical" -q -s > gpurun_out/${tag}_t.log 2>&1; echo "pytest rc=$?"; grep -h "workgroups replayed\|passed\|failed\|Error" gpurun_out/${tag}_t.log | tail -12
timeout 300 python3 bench.py --steps 3 --warmup 1 --side-steps 0 --other-steps 1 --no-cpu-baseline --reference-seconds 0 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/${tag}_bench.json") if l.startswith("{")][-1])
print("wordpress nodes/s %.4e props/s %.4e evals/node %s  %s" % (d["nodes_per_sec"], d["value"], d["evaluations_per_node"], d["config"]["workload"][60:200]))
for o in d.get("other_workloads", []):
    print("  %-28s %-5s nodes/s %.4e props/s %.4e  %d x %d %s  roof %s %.3f" % (o["workload"][:28], o["fixpoint"], o["nodes_per_sec"], o["propagations_per_sec"], o["workgroups"], o["threads"], o["memory"], o["roofline"]["bound"], o["roofline"]["frac"]))
PY
