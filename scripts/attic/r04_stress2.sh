#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for i in $(seq 1 10); do
  timeout 600 python3 -m pytest "tests/test_gpu_parity.py::test_element_models_tree_identical" -q > gpurun_out/r04_stress2_$i.log 2>&1
  echo "both $i: $(tail -1 gpurun_out/r04_stress2_$i.log) $(grep -h 'AssertionError: (' gpurun_out/r04_stress2_$i.log | tr '\n' ' ')"
done
timeout 900 python3 -m pytest tests/test_gpu_fullgrid_paths.py tests/test_headline_trees.py -x -q -s > gpurun_out/r04_fullgrid_t.log 2>&1; echo "fullgrid rc=$?"; grep -h "workgroups replayed\|passed\|failed\|Error" gpurun_out/r04_fullgrid_t.log | head -20
