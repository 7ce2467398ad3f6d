#!/bin/bash
# same-box A/B of a pack-time switch read from the environment (one library): scripts/r06_env_ab.sh VAR  -- headline step, trains15, accap_a3, the proof search; two passes
cd $GRAFT_REPO_ROOT
var=$1
for pass in 1 2; do for val in 0 1; do
  export $var=$val
  timeout 300 python3 bench.py --steps 3 --warmup 1 --side-steps 0 --other-steps 0 --sharded-search 0 --no-cpu-baseline --reference-seconds 0 > /tmp/ab.json 2>/tmp/ab.err
  python3 -c "
import json; d=json.load(open('/tmp/ab.json')); print('$var=$val pass $pass: wordpress7_500 nodes/s %.4e  props/s %.4e  ms/step %.1f' % (d['nodes_per_sec'], d['value'], d['ms_per_step']))"
  for w in trains15 accap_a3; do echo -n "$var=$val pass $pass: "; timeout 200 python3 scripts/quick_rate.py $w nodes=24000000 fixpoint=2 reps=3 2>&1 | tail -1 | cut -c1-80; done
  timeout 200 python3 scripts/valu_by_phase.py 0x0 wordpress7_500 0 proof:500:21 2>&1 | tail -1 | sed "s#^#$var=$val pass $pass proof: #"
done; done
