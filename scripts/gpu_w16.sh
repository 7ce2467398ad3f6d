#!/bin/bash
# wordpress7_500: COMPACT16 and narrower workgroups, same box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for a in "" "debug=0x10100000" "debug=0x10100000 threads_per_block=128 bpc=12" "debug=0x10100000 threads_per_block=128 bpc=10" "debug=0x10100000 threads_per_block=128 bpc=8" "threads_per_block=128 bpc=8" "debug=0x10100000 threads_per_block=64 bpc=12" ""; do
  timeout 200 python3 scripts/quick_rate.py wordpress7_500 nodes=48000000 fixpoint=2 $a 2>&1 | tail -1
done
