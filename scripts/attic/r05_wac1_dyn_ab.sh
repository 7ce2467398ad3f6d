#!/bin/bash
# Slices on demand (kernels.hpp: fixpoint, DYN) in the WAC1 sweeps of the LDS-resident 1024-thread kernels too (-DTB_DYNAMIC_DEEP=1): same-box A/B on the three instances
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $root
for round in 1 2; do for lib in $1; do for w in wordpress7_500 accap_a3 trains15; do
  TURBO_HIP_LIB=$root/turbo_amd/lib/$lib timeout 300 python3 bench.py --workload $w --fixpoint wac1 --steps 2 --warmup 1 --side-steps 0 --other-steps 0 --reference-seconds 0 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib $w wac1 round $round: %.3e nodes/s  %.3e propagations/s' % (d['nodes_per_sec'], d['value']))"
done; done; done
