#!/bin/bash
# same-box A/B over the three headline instances: scripts/gpu_ab3.sh lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for wl in accap_a3 trains15 wordpress7_500; do
for lib in "$@"; do
  TURBO_HIP_LIB=$GRAFT_REPO_ROOT/turbo_amd/lib/$lib timeout 300 python3 bench.py --workload $wl --steps 2 --warmup 1 --side-steps 0 --no-cpu-baseline --reference-seconds 0 > /tmp/ab.json 2>/tmp/ab.err
  python3 -c "
import json; d=json.load(open('/tmp/ab.json')); print('$wl $lib: nodes/s %.4e  props/s %.4e  ms/step %.1f' % (d['nodes_per_sec'], d['value'], d['ms_per_step']))"
done; done
