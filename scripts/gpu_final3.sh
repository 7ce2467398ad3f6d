#!/bin/bash
# new tests, then the round-3 profile of the final kernels (kernel trace + counters), rates, bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python3 -m pytest tests/test_gpu_constants.py -x -q > gpurun_out/final3_t.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/final3_t.log
bash scripts/profile_r03.sh r03 > gpurun_out/final3_profile.log 2>&1; echo "profile rc=$?"
timeout 900 python3 scripts/rates_table.py --out gpurun_out/r03_rates.json > gpurun_out/final3_rates.log 2>&1; echo "rates rc=$?"; grep -c "" gpurun_out/final3_rates.log
timeout 500 python3 bench.py > gpurun_out/r03_bench_final.json 2> gpurun_out/r03_bench_final.err; echo "bench rc=$?"; head -c 600 gpurun_out/r03_bench_final.json
