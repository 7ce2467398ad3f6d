#!/bin/bash
# same-box A/B of engine builds on the synthetic 100k x 500k network: scripts/r06_ab_synth.sh <fixpoint> lib...   (two interleaved passes)
cd $GRAFT_REPO_ROOT
fp=$1; shift
for pass in 1 2; do for lib in "$@"; do
  export TURBO_HIP_LIB=$GRAFT_REPO_ROOT/turbo_amd/lib/$lib
  timeout 300 python3 bench.py --workload synthetic --fixpoint $fp --steps 3 --warmup 1 --side-steps 0 --other-steps 0 --sharded-search 0 --no-cpu-baseline --reference-seconds 0 > /tmp/ab.json 2>/tmp/ab.err
  python3 -c "
import json; d=json.load(open('/tmp/ab.json')); print('$lib pass $pass: synthetic $fp nodes/s %.4e  props/s %.4e  ms/step %.1f' % (d['nodes_per_sec'], d['value'], d['ms_per_step']))"
done; done
