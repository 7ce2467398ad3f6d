#!/bin/bash
# Does the slice hand-out compiled into the LDS-resident 1024-thread sweep kernels (TB_DYNAMIC_DEEP) cost their plain (AC1) sweeps anything?  Same-box A/B, wordpress7_500.
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $root
for round in 1 2; do for lib in $1; do for fp in ac1 wac1; do
  TURBO_HIP_LIB=$root/turbo_amd/lib/$lib timeout 300 python3 bench.py --fixpoint $fp --steps 2 --warmup 1 --side-steps 0 --other-steps 0 --reference-seconds 0 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib wordpress7_500 $fp round $round: %.4e nodes/s  %.4e propagations/s' % (d['nodes_per_sec'], d['value']))"
done; done; done
