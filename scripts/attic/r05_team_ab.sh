#!/bin/bash
# Workgroup teams (kernels.hpp layout 5) against the hot tier on the synthetic 100k x 500k network (BASELINE.json configs[4]), same box:
#   scripts/r05_team_ab.sh           rates of both, then the memory-side counters of the team kernel (L2 hit rate, fabric bytes per propagation)
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $root
out=gpurun_out/team_ab; rm -rf $out; mkdir -p $out
args="--workload synthetic --steps 2 --warmup 1 --side-steps 0 --other-steps 0 --reference-seconds 0 --no-cpu-baseline"
# configurations: TB_TEAM TB_TEAM_SPLIT TB_TEAM_RELAXED (default of this script: the hot tier, then teams of an XCD, relaxed barrier, two and four teams per XCD)
for fp in ${TEAM_FPS:-wac1}; do
  cfgs=${TEAM_CFGS:-0:1:0 1:1:0 1:1:1 1:2:1 1:4:1 0:1:0 1:1:1 1:2:1 1:4:1}
  for cfg in $cfgs; do
    IFS=: read t sp rl <<< "$cfg"
    TB_TEAM=$t TB_TEAM_SPLIT=$sp TB_TEAM_RELAXED=$rl timeout 300 python3 bench.py $args --fixpoint $fp > $out/rate_${fp}_$cfg.log 2>$out/rate_${fp}_$cfg.err
    python3 - <<PY
import json
try:
    d=[json.loads(l) for l in open("$out/rate_${fp}_$cfg.log") if l.startswith("{")][-1]
    print("team=$t split=$sp relaxed=$rl $fp: %.3e propagations/s  %.3e nodes/s" % (d["value"], d["nodes_per_sec"]))
except Exception as e:
    print("team=$t split=$sp relaxed=$rl $fp: failed", e, open("$out/rate_${fp}_$cfg.err").read()[-400:])
PY
  done
done
[ -n "$TEAM_NO_PMC" ] && exit 0
export TB_TEAM_SPLIT=${TEAM_PMC_SPLIT:-1} TB_TEAM_RELAXED=${TEAM_PMC_RELAXED:-0}
pass() { d=$1; shift; c=""; while [ "$1" != "--" ]; do c="$c $1"; shift; done; shift
  TB_TEAM=1 timeout 420 rocprofv3 --pmc $c -d $out/$d -o p -- python3 bench.py "$@" > $out/$d.log 2>&1 || echo "pass $d: rc=$?"; }
pass tcc TCC_HIT_sum TCC_MISS_sum -- $args --fixpoint wac1
pass fetch FETCH_SIZE -- $args --fixpoint wac1
pass write WRITE_SIZE -- $args --fixpoint wac1
pass ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -- $args --fixpoint wac1
python3 - <<PY
import glob, sqlite3, json
res={}
for d in ("tcc","fetch","write","ea"):
    dbs=glob.glob("$out/%s/**/*_results.db" % d, recursive=True)
    if not dbs: continue
    con=sqlite3.connect(dbs[0])
    for k,v,n in con.execute("select counter_name, sum(value), count(distinct dispatch_id) from counters_collection where kernel_name like '%solve_kernel_team%' group by counter_name"):
        res[k]={"sum":v,"launches":n}
    try:
        line=[json.loads(l) for l in open("$out/%s.log" % d) if l.startswith("{")][-1]
        res[d+"_propagations_per_sec"]=line["value"]; res[d+"_ms_per_step"]=line["ms_per_step"]
        res[d+"_props_per_launch"]=line["value"]*line["ms_per_step"]/1000.0
    except Exception as e: res[d+"_line"]=str(e)
if "TCC_HIT_sum" in res: res["l2_hit_rate"]=res["TCC_HIT_sum"]["sum"]/(res["TCC_HIT_sum"]["sum"]+res["TCC_MISS_sum"]["sum"])
if "FETCH_SIZE" in res and "fetch_props_per_launch" in res:
    res["fetch_bytes_per_propagation_raw_kb_x1024"]=res["FETCH_SIZE"]["sum"]*1024/res["FETCH_SIZE"]["launches"]/res["fetch_props_per_launch"]
json.dump(res,open("$out/summary.json","w"),indent=1); print(json.dumps(res))
PY
