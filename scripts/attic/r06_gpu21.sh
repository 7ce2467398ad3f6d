#!/bin/bash
# one pass of the headline / trains15 / accap_a3 rates per library variant (compile-flag lottery)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for lib in ab/base.so $(ls turbo_amd/lib/ab/f_*.so | sed 's#turbo_amd/lib/##') ab/base.so; do
  export TURBO_HIP_LIB=$GRAFT_REPO_ROOT/turbo_amd/lib/$lib
  timeout 300 python3 bench.py --steps 3 --warmup 1 --side-steps 0 --other-steps 0 --sharded-search 0 --no-cpu-baseline --reference-seconds 0 > /tmp/ab.json 2>/tmp/ab.err
  python3 -c "
import json; d=json.load(open('/tmp/ab.json')); print('$lib: wordpress7_500 nodes/s %.4e  ms/step %.1f' % (d['nodes_per_sec'], d['ms_per_step']))"
  for w in trains15 accap_a3; do echo -n "$lib: "; timeout 200 python3 scripts/quick_rate.py $w nodes=24000000 fixpoint=2 reps=3 2>&1 | tail -1 | cut -c1-60; done
done > gpurun_out/r06_flags.txt 2>&1
cat gpurun_out/r06_flags.txt
