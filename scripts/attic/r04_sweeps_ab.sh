#!/bin/bash
# same-box A/B of the sweeping kernels (-fp wac1 / ac1): in-tree library against turbo_amd/lib/ab/$1.so, all BASELINE configurations
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
v=${1:-pf3}
for w in wordpress7_500 accap_a3 trains15 synthetic; do for fp in wac1 ac1; do for lib in libturbo_hip.so ab/$v.so; do
  [ $w = synthetic ] && [ $fp = ac1 ] && continue
  TURBO_HIP_LIB=$GRAFT_REPO_ROOT/turbo_amd/lib/$lib timeout 300 python3 bench.py --workload $w --fixpoint $fp --steps 2 --warmup 1 --side-steps 0 --other-steps 0 --no-cpu-baseline --reference-seconds 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])
print(\"$w $fp $lib: nodes/s %.4e props/s %.4e\" % (d[\"nodes_per_sec\"], d[\"value\"]))"
done; done; done
