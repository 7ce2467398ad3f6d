#!/bin/bash
# the N = 4 and N = 8 paths of bench.py end to end on a 1-GPU box (--share-device: every rank takes 1/N of the CUs): handles of 8 processes, 56 peer mappings, stealing among 8
cd $GRAFT_REPO_ROOT
for n in 4 8; do
  timeout 600 python3 bench.py --gpus $n --share-device --steps 2 --warmup 1 --side-steps 0 --other-steps 0 --reference-seconds 0 --no-cpu-baseline > gpurun_out/r06_world$n.json 2> gpurun_out/r06_world$n.err
  echo "N=$n rc=$?"; tail -3 gpurun_out/r06_world$n.err | cut -c1-300
  python3 -c "
import json; d=json.loads(open('gpurun_out/r06_world$n.json').read().strip().splitlines()[-1]); s=d.get('sharded_search',{})
print({k:d.get(k) for k in ('n_gpus','value','nodes_per_sec','ms_per_step','scaling')}); print({k:s.get(k) for k in ('seconds','linked','nodes','eps_solved','eps_skipped','stolen_subproblems','every_subproblem_accounted_once')})
print([ (r.get('rank'), r.get('eps_solved'), r.get('stolen'), round(r.get('wait_share',0),3)) for r in s.get('per_rank',[])])"
done
