cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TURBO_HIP_LIB=$GRAFT_REPO_ROOT/turbo_amd/lib/libturbo_hip_tuning.so TB_DUMP_SLICES=1 TB_CENSUS_ROWS=120 timeout 300 python3 scripts/r04_slice_census.py example_wordpress7_500.fzn > gpurun_out/r06_census_wp.log 2>&1
grep -c "slice-census" gpurun_out/r06_census_wp.log; tail -3 gpurun_out/r06_census_wp.log
