#!/bin/bash
# final: profile of the bench command (trace + counters), bench line, full GPU suite
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash scripts/profile_r03.sh r03 > gpurun_out/final4_profile.log 2>&1; echo "profile rc=$?"
timeout 600 python3 bench.py > gpurun_out/r03_bench_final.json 2> gpurun_out/r03_bench_final.err; echo "bench rc=$?"; head -c 400 gpurun_out/r03_bench_final.json; echo
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/final4_t.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/final4_t.log
