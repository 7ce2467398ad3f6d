#!/bin/bash
# GLOBAL store (synthetic 100k x 500k): variables renumbered in first-touch order of the propagator stream (TB_GLOBAL_RENUMBER) against the caller's numbering
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export TURBO_HIP_LIB=${1:-turbo_amd/lib/libturbo_hip.so}
for fp in wac1 event; do
  for v in caller renumbered; do
    unset TB_GLOBAL_RENUMBER; [ $v = renumbered ] && export TB_GLOBAL_RENUMBER=1
    timeout 300 python3 bench.py --workload synthetic --fixpoint $fp --steps 2 --warmup 1 --side-steps 0 --other-steps 0 --no-cpu-baseline --reference-seconds 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$fp $v: propagations/s %.4e nodes/s %.4e evals/node %.0f' % (d['value'], d['nodes_per_sec'], d['value'] / d['nodes_per_sec']))"
  done
done
