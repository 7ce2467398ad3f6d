#!/bin/bash
# fresh seeds beyond those of the suite on the production library of the round (the oracle is the checker): node-level network fuzz, element-model trees, full-occupancy copies
cd $GRAFT_REPO_ROOT
timeout 1500 python3 tests/tools/fuzz_long.py ${1:-20000} ${2:-1200} 2>&1 | tail -3 > gpurun_out/r06_fuzz_long.log; cat gpurun_out/r06_fuzz_long.log
timeout 1500 python3 tests/tools/stress_element.py ${3:-9000} ${4:-64} 2>&1 | tail -2 > gpurun_out/r06_stress_element.log; cat gpurun_out/r06_stress_element.log
timeout 600 python3 tests/tools/stress_event.py test_data/sudoku_opt4.fzn test_data/pat2.fzn accap_a3.fzn trains15.fzn 2>&1 | tail -4 > gpurun_out/r06_stress_event.log; cat gpurun_out/r06_stress_event.log
