#!/bin/bash
# constants out of the slab: parity subset first, then wordpress7_500 at several workgroup shapes (same box)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_headline_trees.py tests/test_gpu_selfcheck.py -x -q > gpurun_out/outs_t.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/outs_t.log
for a in "" "debug=0x40000000" "threads_per_block=128 bpc=13" "threads_per_block=128 bpc=12" "threads_per_block=128 bpc=10" "threads_per_block=128 bpc=8" "bpc=8" ""; do
  timeout 200 python3 scripts/quick_rate.py wordpress7_500 nodes=48000000 fixpoint=2 $a 2>&1 | tail -1
done
for w in trains15 accap_a3; do for a in "" "debug=0x40000000"; do
  timeout 200 python3 scripts/quick_rate.py $w nodes=24000000 fixpoint=2 $a 2>&1 | tail -1
done; done
