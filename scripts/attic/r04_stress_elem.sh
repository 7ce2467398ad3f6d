#!/bin/bash
# flakiness hunt: the element-model fuzz repeatedly under each combination of the wake-up filters
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
n=${1:-5}
for v in none cond range both; do
  unset TB_NO_COND_WAKE TB_NO_CHAIN_RANGE
  case $v in none) export TB_NO_COND_WAKE=1 TB_NO_CHAIN_RANGE=1;; cond) export TB_NO_CHAIN_RANGE=1;; range) export TB_NO_COND_WAKE=1;; esac
  for i in $(seq 1 $n); do
    timeout 600 python3 -m pytest "tests/test_gpu_parity.py::test_element_models_tree_identical" -q > gpurun_out/r04_stress_${v}_$i.log 2>&1
    echo "$v $i: $(tail -1 gpurun_out/r04_stress_${v}_$i.log) $(grep -h 'AssertionError: (' gpurun_out/r04_stress_${v}_$i.log | tr '\n' ' ')"
  done
done
