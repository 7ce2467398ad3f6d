cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for ev in 0 1; do
  for sp in 4 2 8; do
    [ $ev = 0 ] && [ $sp != 4 ] && continue
    TB_TEAM_EVENT=$ev TB_TEAM_SPLIT=$sp timeout 300 python3 bench.py --workload synthetic --fixpoint event --steps 3 --warmup 1 --side-steps 0 --other-steps 0 --no-cpu-baseline --sharded-search 0 > /tmp/b.json 2>/tmp/b.err
    python3 -c "
import json; d=json.load(open('/tmp/b.json')); print('TB_TEAM_EVENT=$ev split $sp: nodes/s %.4e props/s %.4e ms/step %.1f  %s' % (d['nodes_per_sec'], d['value'], d['ms_per_step'], d['config']['workload'][-120:]))" || tail -3 /tmp/b.err
  done
done
timeout 600 python3 -m pytest tests/test_gpu_fullsize_global.py -x -q -s > gpurun_out/r06h_t.log 2>&1; echo "pytest fullsize rc=$?"; tail -8 gpurun_out/r06h_t.log
