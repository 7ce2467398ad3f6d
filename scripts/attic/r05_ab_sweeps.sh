#!/bin/bash
# same-box A/B of engine builds on the SWEEPING fixpoints and the synthetic network: scripts/r05_ab_sweeps.sh lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT
common="--steps 2 --warmup 1 --side-steps 0 --other-steps 0 --reference-seconds 0 --no-cpu-baseline"
for lib in "$@"; do
  name=$(basename $lib .so)
  for cfg in "wordpress7_500 wac1" "wordpress7_500 ac1" "trains15 wac1" "synthetic wac1" "synthetic event" "synthetic ac1"; do
    set -- $cfg
    TURBO_HIP_LIB=$lib timeout 300 python3 bench.py $common --workload $1 --fixpoint $2 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$name $1 $2: nodes/s %.4e  propagations/s %.4e' % (d['nodes_per_sec'], d['value']))
"
  done
done
