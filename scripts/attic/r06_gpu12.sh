cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python3 -m pytest tests/test_headline_trees.py tests/test_gpu_selfcheck.py tests/test_gpu_fullgrid_paths.py -x -q > gpurun_out/r06f_t.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r06f_t.log
AB_NO_PMC= bash scripts/r06_ab.sh ab/v2.so ab/v3.so > gpurun_out/r06_ab_v2_v3.txt 2>&1; cat gpurun_out/r06_ab_v2_v3.txt
