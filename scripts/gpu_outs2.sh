#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for a in "" "debug=0x40000000" "bpc=12" "bpc=10" ""; do
  timeout 200 python3 scripts/quick_rate.py wordpress7_500 nodes=48000000 fixpoint=2 $a 2>&1 | tail -1
done
for w in trains15 accap_a3; do for a in "" "fixpoint=1"; do
  timeout 200 python3 scripts/quick_rate.py $w nodes=24000000 fixpoint=2 $a 2>&1 | tail -1
done; done
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/outs2_t.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/outs2_t.log
