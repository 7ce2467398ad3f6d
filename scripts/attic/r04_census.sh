#!/bin/bash
# round 4, first GPU call: census of the event fixpoint on wordpress7_500 with the tuning build (what runs, what narrows nothing, slice classes)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export TURBO_HIP_LIB=turbo_amd/lib/libturbo_hip_tuning.so
timeout 300 python3 scripts/event_profile_gpu.py > gpurun_out/r04_event_profile.log 2>&1; echo "event_profile rc=$?"
timeout 600 python3 scripts/useless_runs_probe.py > gpurun_out/r04_useless.log 2>&1; echo "useless rc=$?"
TB_DUMP_SLICES=1 timeout 300 python3 - > gpurun_out/r04_slices.log 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from turbo_amd import capi, preprocess
_, tcn, _ = preprocess.load_fzn_simplified("benchmarks/example_wordpress7_500.fzn")
cfg = capi.make_config(fixpoint=2, stop_after_n_nodes_total=100000, timeout_ms=60000)
s = capi.Session(tcn, cfg); print(s.plan()); s.close()
PY
echo "slices rc=$?"
tail -12 gpurun_out/r04_event_profile.log; cat gpurun_out/r04_useless.log
