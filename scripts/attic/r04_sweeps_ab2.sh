#!/bin/bash
# wac1 only, the 1024-thread configurations: in-tree library against turbo_amd/lib/ab/$1.so, twice
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
v=${1:-pf5}
for round in 1 2; do for w in wordpress7_500 synthetic; do for lib in libturbo_hip.so ab/$v.so; do
  TURBO_HIP_LIB=$GRAFT_REPO_ROOT/turbo_amd/lib/$lib timeout 300 python3 bench.py --workload $w --fixpoint wac1 --steps 2 --warmup 1 --side-steps 0 --other-steps 0 --no-cpu-baseline --reference-seconds 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])
print(\"$w wac1 $lib: nodes/s %.4e props/s %.4e\" % (d[\"nodes_per_sec\"], d[\"value\"]))"
done; done; done
