#!/bin/bash
# same-box, same-library A/B through an environment switch of the host shim: scripts/gpu_env_ab.sh VAR workload [workload ...]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
var=$1; shift
for wl in "$@"; do
for on in 0 1 0 1; do
  if [ $on = 1 ]; then export $var=1; else unset $var; fi
  timeout 300 python3 bench.py --workload $wl --steps 2 --warmup 1 --side-steps 0 --no-cpu-baseline --reference-seconds 0 > /tmp/ab.json 2>/tmp/ab.err
  python3 -c "
import json; d=json.load(open('/tmp/ab.json')); print('$wl $var=$on: nodes/s %.4e  props/s %.4e  ms/step %.1f  %s' % (d['nodes_per_sec'], d['value'], d['ms_per_step'], d['config']['workload'].split(',')[1:3]))"
done; done
