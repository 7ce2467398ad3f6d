#!/bin/bash
# first GPU call of round 3: test suite, bench line, phase budget of the r02 kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03_t1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_t1.log
tail -5 gpurun_out/r03_t1.log
timeout 400 python3 bench.py > gpurun_out/r03_bench0.json 2> gpurun_out/r03_bench0.err; echo "bench rc=$?"
head -c 1500 gpurun_out/r03_bench0.json
timeout 600 python3 scripts/phase_budget.py r03a wordpress7_500 12000000 > gpurun_out/r03a_phase.log 2>&1; echo "phase rc=$?"
tail -60 gpurun_out/r03a_phase.log
