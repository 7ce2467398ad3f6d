#!/bin/bash
# same-box A/B of engine builds: scripts/gpu_ab.sh lib1.so lib2.so ...  (two interleaved passes each)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for pass in 1 2; do
for lib in "$@"; do
  TURBO_HIP_LIB=$GRAFT_REPO_ROOT/turbo_amd/lib/$lib timeout 300 python3 bench.py --steps 3 --warmup 1 --side-steps 0 --no-cpu-baseline --reference-seconds 0 ${AB_ARGS} > /tmp/ab.json 2>/tmp/ab.err
  python3 -c "
import json; d=json.load(open('/tmp/ab.json')); print('$lib pass $pass: nodes/s %.4e  props/s %.4e  ms/step %.1f' % (d['nodes_per_sec'], d['value'], d['ms_per_step']))"
done; done
