#!/bin/bash
# a five-minute check of a tree on the GPU: smoke(), the parity core, the two-rank path, the bench contract line
cd $GRAFT_REPO_ROOT
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullgrid_paths.py tests/test_gpu_fullsize_global.py tests/test_gpu_multi.py tests/test_gpu_cli.py -m gpu -x -q 2>&1 | tail -3
