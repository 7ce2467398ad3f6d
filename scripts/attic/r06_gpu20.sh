#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scripts/r06_ab.sh ab/chan0.so ab/chan1.so > gpurun_out/r06_ab_chan.txt 2>&1
tail -20 gpurun_out/r06_ab_chan.txt
timeout 900 python3 -m pytest tests/test_gpu_team.py -q -x -k "racing or real_xcds" 2>&1 | tail -5
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullgrid_paths.py -q -x 2>&1 | tail -3
