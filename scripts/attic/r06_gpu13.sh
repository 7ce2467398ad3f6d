cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_team.py -x -q -k "event" > gpurun_out/r06g_t.log 2>&1; echo "pytest event teams rc=$?"; tail -12 gpurun_out/r06g_t.log
