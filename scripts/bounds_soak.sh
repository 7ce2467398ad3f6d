#!/bin/bash
# GPU box: everything that runs on the software bounds build (make bounds; kernels.hpp TB_BOUNDS).  usage: scripts/bounds_soak.sh [seconds per search configuration]
#  1. scripts/bounds_soak.py: the bench workloads and the layouts the planner does not pick by itself, full grids, a wall-clock budget each
#  2. the GPU test files that drive the engine through ctypes (fuzz families, two-rank suite, depth, streaming, leaf rules, orders) with TURBO_HIP_LIB pointing at the bounds library
#  3. tests/tools/stress_element.py on fresh seeds
#    (not test_a_grid_that_never_becomes_resident...: it runs two sessions of DIFFERENT networks at once, and the limits of the bounds build are per-process device
#     variables written before each launch -- the second session's limits would judge the first one's indexes: a false report, r06)
# A violation surfaces as TB_ERR_HIP "bounds build: ... site S, index I, limit L, workgroup W" in whichever step hits it.
cd "$(dirname "$0")/.."
export TURBO_HIP_LIB=$PWD/turbo_amd/lib/libturbo_hip_bounds.so
mkdir -p gpurun_out
timeout 2400 python3 scripts/bounds_soak.py "${1:-45}" gpurun_out/r06_bounds_soak.json > gpurun_out/r06_bounds_soak.log 2>&1
echo "soak rc=$?" >> gpurun_out/r06_bounds_soak.log
timeout 2400 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_compact8.py tests/test_gpu_hot_tier.py tests/test_gpu_leaf_rule.py tests/test_gpu_orders.py \
  tests/test_gpu_depth.py tests/test_gpu_constants.py tests/test_gpu_streaming.py tests/test_gpu_fullgrid_paths.py tests/test_headline_trees.py tests/test_gpu_watchdog.py tests/test_gpu_team.py tests/test_gpu_fullsize_global.py \
  -m gpu -q -x --deselect tests/test_gpu_leaf_rule.py::test_cli_arch_selects_the_leaf_rule --deselect tests/test_gpu_orders.py::test_cli_eps_orders_walk_the_oracles_tree \
  --deselect tests/test_gpu_team.py::test_a_grid_that_never_becomes_resident_is_reported_not_waited_for 2>&1 | tail -15 > gpurun_out/r06_bounds_pytest.log
timeout 1200 python3 tests/tools/stress_element.py 7000 48 2>&1 | tail -5 > gpurun_out/r06_bounds_stress.log
grep -h "bounds build" gpurun_out/r06_bounds_soak.log gpurun_out/r06_bounds_pytest.log gpurun_out/r06_bounds_stress.log | head -5
tail -3 gpurun_out/r06_bounds_soak.log; tail -3 gpurun_out/r06_bounds_pytest.log; tail -2 gpurun_out/r06_bounds_stress.log
