#!/bin/bash
# full GPU suite + default bench line on the current tree
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03_t3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_t3.log
tail -5 gpurun_out/r03_t3.log
timeout 500 python3 bench.py > gpurun_out/r03_bench3.json 2> gpurun_out/r03_bench3.err; echo "bench rc=$?"
head -c 3000 gpurun_out/r03_bench3.json
tail -3 gpurun_out/r03_bench3.err
