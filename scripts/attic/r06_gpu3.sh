cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
# (1) does the instrumented wordpress7_500 kernel fault at every size / is it the kernel or the instance?
for n in 200000 2000000; do
  TURBO_HIP_LIB=$GRAFT_REPO_ROOT/turbo_amd/lib/libturbo_hip_blocks.so TB_BLOCK_COUNTS=$GRAFT_REPO_ROOT/gpurun_out/r06_probe.bin timeout 300 python3 scripts/valu_by_phase.py 0x0 wordpress7_500 $n 2>&1 | tail -2
done
echo "accap_a3 forced COMPACT (the wordpress kernel on another instance):"
TURBO_HIP_LIB=$GRAFT_REPO_ROOT/turbo_amd/lib/libturbo_hip_blocks.so TB_BLOCK_COUNTS=$GRAFT_REPO_ROOT/gpurun_out/r06_probe.bin timeout 300 python3 scripts/valu_by_phase.py 0x100000 accap_a3 2000000 2>&1 | tail -2
echo "wordpress7_500 never COMPACT (the accap kernel on wordpress):"
TURBO_HIP_LIB=$GRAFT_REPO_ROOT/turbo_amd/lib/libturbo_hip_blocks.so TB_BLOCK_COUNTS=$GRAFT_REPO_ROOT/gpurun_out/r06_probe.bin timeout 300 python3 scripts/valu_by_phase.py 0x80000 wordpress7_500 2000000 2>&1 | tail -2
# (2) the budget
timeout 1500 bash scripts/r06_blocks.sh 12000000 > gpurun_out/r06_blocks.log 2>&1; echo "blocks rc=$?"; grep -v "^  \|^   " gpurun_out/r06_blocks.log | tail -30
# (3) tests
timeout 300 python3 -m pytest tests/test_gpu_team.py -x -q -k "never_becomes_resident" > gpurun_out/r06b_t.log 2>&1; echo "pytest team rc=$?"; tail -15 gpurun_out/r06b_t.log
timeout 600 python3 -m pytest tests/test_gpu_fullsize_global.py -x -q -s --durations=5 > gpurun_out/r06c_t.log 2>&1; echo "pytest fullsize rc=$?"; tail -25 gpurun_out/r06c_t.log
timeout 900 python3 -m pytest tests/test_gpu_multi.py -x -q --durations=8 > gpurun_out/r06d_t.log 2>&1; echo "pytest multi rc=$?"; tail -25 gpurun_out/r06d_t.log
