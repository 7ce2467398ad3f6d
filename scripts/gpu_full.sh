#!/bin/bash
# The GPU suite as the driver runs it (`pytest -m gpu`: every family, wide instance x mode sweeps thinned -- tests/conftest.py), with the slowest tests listed, and -- with
# "soak" as first argument -- the rest of the cross products too (`pytest -m "gpu and soak"`).  Then the three headline rates.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests -m gpu -x -q --durations=40 > gpurun_out/full_t.log 2>&1; echo "pytest -m gpu rc=$?"; tail -4 gpurun_out/full_t.log
grep -A45 "slowest 40 durations" gpurun_out/full_t.log > gpurun_out/full_durations.txt
if [ "$1" = "soak" ]; then
  timeout 1500 python3 -m pytest tests -m "gpu and soak" -x -q --durations=20 > gpurun_out/soak_t.log 2>&1; echo "pytest -m 'gpu and soak' rc=$?"; tail -4 gpurun_out/soak_t.log
fi
for w in wordpress7_500 trains15 accap_a3; do timeout 200 python3 scripts/quick_rate.py $w nodes=48000000 fixpoint=2 2>&1 | tail -1; done
