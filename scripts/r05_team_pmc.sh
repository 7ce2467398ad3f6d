#!/bin/bash
# What bounds a team's sweeps on the synthetic 100k x 500k network?  Memory-pipeline counters of solve_kernel_team (separate --pmc passes, no tracing):
#   scripts/r05_team_pmc.sh [wac1|ac1]
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $root
fp=${1:-wac1}
out=gpurun_out/team_pmc_$fp; rm -rf $out; mkdir -p $out
args="--workload synthetic --steps 2 --warmup 1 --side-steps 0 --other-steps 0 --reference-seconds 0 --no-cpu-baseline --fixpoint $fp"
pass() { d=$1; shift; timeout 420 rocprofv3 --pmc "$@" -d $out/$d -o p -- python3 bench.py $args > $out/$d.log 2>&1 || echo "pass $d: rc=$?"; }
pass lat TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
pass atom TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum
pass tcc TCC_REQ_sum TCC_BUSY_sum TCC_TAG_STALL_sum
pass tcp TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS
pass act SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
pass grbm GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum
python3 - <<PY
import glob, sqlite3, json
res={}
for d in ("lat","atom","tcc","tcp","sq","insts","act","grbm"):
    dbs=glob.glob("$out/%s/**/*_results.db" % d, recursive=True)
    if not dbs: res[d]="no db"; continue
    con=sqlite3.connect(dbs[0])
    tabs=[r[0] for r in con.execute("select name from sqlite_master where type in ('view','table')")]
    try:
        for k,v,n in con.execute("select counter_name, sum(value), count(distinct dispatch_id) from counters_collection where kernel_name like '%solve_kernel_team%' group by counter_name"):
            res[k]=v/max(n,1)
    except Exception as e: res[d]=str(e)[:200]
    try:
        line=[json.loads(l) for l in open("$out/%s.log" % d) if l.startswith("{")][-1]
        res[d+"_ms_per_launch"]=line["ms_per_step"]; res[d+"_props_per_launch"]=line["value"]*line["ms_per_step"]/1000.0
    except Exception as e: res[d+"_line"]=str(e)[:100]
json.dump(res,open("$out/summary.json","w"),indent=1); print(json.dumps(res, indent=1))
PY
