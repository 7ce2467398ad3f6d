#!/bin/bash
# the whole GPU suite, then the three headline rates
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1800 python3 -m pytest tests -m gpu -x -q > gpurun_out/full_t.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/full_t.log
for w in wordpress7_500 trains15 accap_a3; do timeout 200 python3 scripts/quick_rate.py $w nodes=48000000 fixpoint=2 2>&1 | tail -1; done
timeout 200 python3 scripts/quick_rate.py accap_a3 nodes=24000000 fixpoint=1 2>&1 | tail -1
