#!/bin/bash
# the round's final evidence in one call: GPU suite with durations, the default bench line, trace + counter passes, block budget
cd $GRAFT_REPO_ROOT
bash scripts/gpu_full.sh > gpurun_out/r06_final_suite.log 2>&1; tail -6 gpurun_out/r06_final_suite.log
python3 bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_line.err; tail -c 300 gpurun_out/r06_bench_line.json; echo
bash scripts/profile_counters.sh r06 thos > gpurun_out/r06_profile.log 2>&1; tail -2 gpurun_out/r06_profile.log
bash scripts/r06_blocks.sh 12000000 wordpress7_500_proof wordpress7_500 trains15 accap_a3 > gpurun_out/r06_blocks.log 2>&1; grep -c "valu_plus_lane_over_measured" gpurun_out/r06_blocks.log
