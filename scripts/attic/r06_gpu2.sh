cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 bash scripts/r06_blocks.sh 12000000 > gpurun_out/r06_blocks.log 2>&1; echo "blocks rc=$?"; tail -60 gpurun_out/r06_blocks.log
timeout 300 python3 -m pytest tests/test_gpu_team.py -x -q -k "never_becomes_resident or small_grids" > gpurun_out/r06b_t.log 2>&1; echo "pytest team rc=$?"; tail -15 gpurun_out/r06b_t.log
timeout 600 python3 -m pytest tests/test_gpu_fullsize_global.py -x -q -s --durations=5 > gpurun_out/r06c_t.log 2>&1; echo "pytest fullsize rc=$?"; tail -25 gpurun_out/r06c_t.log
