#!/bin/bash
# round 4: same-box A/B of the two wake-up filters of the event fixpoint (TB_NO_COND_WAKE / TB_NO_CHAIN_RANGE switch them off at pack time),
# correctness first (headline trees, element-model fuzz, self-check of the tuning build), then runs per node (tuning build) and the bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=${1:-r04ab}
timeout 900 python3 -m pytest tests/test_headline_trees.py tests/test_gpu_selfcheck.py "tests/test_gpu_parity.py::test_element_models_tree_identical" "tests/test_gpu_parity.py::test_compact_slab_in_global_memory_with_a_ragged_implication_slice" "tests/test_gpu_parity.py::test_channelling_networks_bit_exact" -x -q > gpurun_out/${tag}_t.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/${tag}_t.log
for v in none cond range both; do
  unset TB_NO_COND_WAKE TB_NO_CHAIN_RANGE
  case $v in none) export TB_NO_COND_WAKE=1 TB_NO_CHAIN_RANGE=1;; cond) export TB_NO_CHAIN_RANGE=1;; range) export TB_NO_COND_WAKE=1;; esac
  TURBO_HIP_LIB=turbo_amd/lib/libturbo_hip_tuning.so timeout 300 python3 scripts/useless_runs_probe.py > gpurun_out/${tag}_runs_$v.log 2>&1
  timeout 300 python3 bench.py --steps 3 --warmup 1 --side-steps 0 --no-cpu-baseline --reference-seconds 0 > gpurun_out/${tag}_bench_$v.json 2> gpurun_out/${tag}_bench_$v.err; echo "bench $v rc=$?"
  python3 - <<PY
import json
d=json.load(open("gpurun_out/${tag}_bench_$v.json"))
print("$v: nodes/s %.4e  props/s %.4e  ms/step %.1f  evals/node %.0f" % (d["nodes_per_sec"], d["value"], d["ms_per_step"], d["value"]/d["nodes_per_sec"]))
PY
  head -4 gpurun_out/${tag}_runs_$v.log
done
