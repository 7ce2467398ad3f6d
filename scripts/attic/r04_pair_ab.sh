#!/bin/bash
# same-box A/B of the paired implication runs (turbo_amd/lib/ab/pair.so against the in-tree library), then parity of the variant
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash scripts/r04_ab_libs.sh r04pair $GRAFT_REPO_ROOT/turbo_amd/lib/libturbo_hip.so $GRAFT_REPO_ROOT/turbo_amd/lib/ab/pair.so
for w in accap_a3 trains15; do for lib in libturbo_hip.so ab/pair.so; do
  TURBO_HIP_LIB=$GRAFT_REPO_ROOT/turbo_amd/lib/$lib timeout 300 python3 bench.py --workload $w --steps 2 --warmup 1 --side-steps 0 --other-steps 0 --no-cpu-baseline --reference-seconds 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])
print(\"$w $lib: nodes/s %.4e evals/node %.0f\" % (d[\"nodes_per_sec\"], d[\"value\"] / d[\"nodes_per_sec\"]))"
done; done
TURBO_HIP_LIB=$GRAFT_REPO_ROOT/turbo_amd/lib/ab/pair.so timeout 900 python3 -m pytest tests/test_headline_trees.py tests/test_gpu_fullgrid_paths.py "tests/test_gpu_parity.py::test_element_models_tree_identical" "tests/test_gpu_parity.py::test_class_pure_finite_networks_bit_exact" tests/test_gpu_selfcheck.py -x -q 2>&1 | tail -4
