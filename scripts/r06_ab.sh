#!/bin/bash
# same-box A/B of engine builds (r06): scripts/r06_ab.sh ab/a.so ab/b.so ...   -- two interleaved passes: headline bench (3 steps), trains15, accap_a3; then one PMC pass each
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for pass in 1 2; do
for lib in "$@"; do
  export TURBO_HIP_LIB=$GRAFT_REPO_ROOT/turbo_amd/lib/$lib
  timeout 300 python3 bench.py --steps 3 --warmup 1 --side-steps 0 --other-steps 0 --sharded-search 0 --no-cpu-baseline --reference-seconds 0 > /tmp/ab.json 2>/tmp/ab.err
  python3 -c "
import json; d=json.load(open('/tmp/ab.json')); print('$lib pass $pass: wordpress7_500 nodes/s %.4e  props/s %.4e  ms/step %.1f' % (d['nodes_per_sec'], d['value'], d['ms_per_step']))"
  for w in trains15 accap_a3; do echo -n "$lib pass $pass: "; timeout 200 python3 scripts/quick_rate.py $w nodes=24000000 fixpoint=2 reps=3 2>&1 | tail -1; done
done; done
if [ -z "$AB_NO_PMC" ]; then
for lib in "$@"; do
  export TURBO_HIP_LIB=$GRAFT_REPO_ROOT/turbo_amd/lib/$lib
  for w in wordpress7_500 trains15 accap_a3; do
    out=/tmp/pmc_$$; rm -rf $out
    (cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH -d $out -o p -- python3 $GRAFT_REPO_ROOT/scripts/valu_by_phase.py 0x0 $w 12000000 > $out.log 2>&1)
    python3 - <<PY
import glob, sqlite3, re
log=open("$out.log").read()
m=re.search(r"nodes=(\d+) fails=(\d+) deductions=(\d+) kernel_ns=(\d+)", log)
n=int(m.group(1)); ns=int(m.group(4))
db=glob.glob("$out/**/*_results.db", recursive=True)[0]
c={k:v for k,v in sqlite3.connect(db).execute("select counter_name, sum(value) from counters_collection where kernel_name like '%solve_kernel%' group by counter_name")}
print("$lib $w pmc: nodes/s %.3e props/node %.0f" % (n/(ns*1e-9), int(m.group(3))/n), {k.replace("SQ_INSTS_",""):round(v/n,1) for k,v in sorted(c.items())})
PY
  done
done
fi
