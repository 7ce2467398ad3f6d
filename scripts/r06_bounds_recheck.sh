#!/bin/bash
# after the last kernel change of the round (sums of the COMPACT kernels in a pass loop of their own): the parity core and fresh fuzz seeds on the software bounds build
cd $GRAFT_REPO_ROOT
export TURBO_HIP_LIB=$PWD/turbo_amd/lib/libturbo_hip_bounds.so
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_headline_trees.py tests/test_gpu_fullgrid_paths.py tests/test_gpu_constants.py -m gpu -q -x 2>&1 | tail -3
timeout 900 python3 tests/tools/fuzz_long.py 50000 3000 2>&1 | tail -2
timeout 600 python3 scripts/bounds_soak.py 20 gpurun_out/r06_bounds_soak2.json 2>&1 | tail -2
unset TURBO_HIP_LIB
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python3 -m pytest tests/test_gpu_cli.py tests/test_gpu_multi.py -m gpu -q -x 2>&1 | tail -2
