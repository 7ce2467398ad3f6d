#!/bin/bash
# same-box A/B of a library built with other backend flags (turbo_amd/lib/ab/$1.so) against the in-tree one: headline, then accap_a3 / trains15
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
v=${1:-trk}
bash scripts/r04_ab_libs.sh r04$v $GRAFT_REPO_ROOT/turbo_amd/lib/libturbo_hip.so $GRAFT_REPO_ROOT/turbo_amd/lib/ab/$v.so
for w in accap_a3 trains15; do for lib in libturbo_hip.so ab/$v.so libturbo_hip.so ab/$v.so; do
  TURBO_HIP_LIB=$GRAFT_REPO_ROOT/turbo_amd/lib/$lib timeout 300 python3 bench.py --workload $w --steps 2 --warmup 1 --side-steps 0 --other-steps 0 --no-cpu-baseline --reference-seconds 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])
print(\"$w $lib: nodes/s %.4e\" % d[\"nodes_per_sec\"])"
done; done
