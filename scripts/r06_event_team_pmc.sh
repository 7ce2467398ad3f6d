#!/bin/bash
# memory side and issue side of the event fixpoint in workgroup teams (TB_TEAM_EVENT=1, synthetic 100k x 500k), separate --pmc passes, next to the hot tier's
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_evteam; rm -rf $out; mkdir -p $out
args="--workload synthetic --fixpoint event --steps 2 --warmup 1 --side-steps 0 --other-steps 0 --reference-seconds 0 --no-cpu-baseline --sharded-search 0"
for mode in team hot; do
  if [ $mode = team ]; then export TB_TEAM_EVENT=1; else unset TB_TEAM_EVENT; fi
  python3 bench.py $args > $out/${mode}_plain.log 2>&1
  for set in "fetch FETCH_SIZE" "tcc TCC_HIT_sum TCC_MISS_sum" "sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "grbm GRBM_GUI_ACTIVE"; do
    set -- $set; name=$1; shift
    timeout 300 rocprofv3 --pmc $@ -d $out/${mode}_$name -o p -- python3 bench.py $args > $out/${mode}_$name.log 2>&1 || echo "pass ${mode}_$name rc=$?"
  done
done
python3 - <<'PY'
import glob, json, sqlite3
out="gpurun_out/prof_evteam"; res={}
def line_of(path):
    try: return json.loads([l for l in open(path) if l.startswith("{")][-1])
    except Exception: return {}
def counters(d):
    db=glob.glob(f"{out}/{d}/**/*_results.db", recursive=True); c={}
    if db:
        for k,v,n in sqlite3.connect(db[0]).execute("select counter_name, sum(value), count(*) from counters_collection where kernel_name like '%solve_kernel%' group by counter_name"): c[k]=v/max(1,n)
    return c
for mode in ("team","hot"):
    b=line_of(f"{out}/{mode}_plain.log")
    r={"nodes_per_sec": b.get("nodes_per_sec"), "propagations_per_sec": b.get("value"), "kernel": {k: b.get("plan", {}).get(k) for k in ("kernel_opt","num_blocks","threads_per_block")} if isinstance(b.get("plan"), dict) else None}
    f, t, q, g = counters(f"{mode}_fetch"), counters(f"{mode}_tcc"), counters(f"{mode}_sq1"), counters(f"{mode}_grbm")
    pf = line_of(f"{out}/{mode}_fetch.log").get("balance", {}).get("propagations")
    if f.get("FETCH_SIZE") and pf: r["fabric_read_bytes_per_propagation"]=f["FETCH_SIZE"]*1024.0/pf
    if t.get("TCC_HIT_sum") is not None: r["tcc_hit_rate"]=t["TCC_HIT_sum"]/max(1.0,t["TCC_HIT_sum"]+t.get("TCC_MISS_sum",0.0))
    bq=line_of(f"{out}/{mode}_sq1.log").get("balance", {})
    if q and bq.get("propagations"):
        r["valu_per_64_propagations"]=q["SQ_INSTS_VALU"]/(bq["propagations"]/64.0); r["salu_per_64_propagations"]=q["SQ_INSTS_SALU"]/(bq["propagations"]/64.0)
        r["wait_any_share"]=q["SQ_WAIT_ANY"]/q["SQ_WAVE_CYCLES"]
        if g.get("GRBM_GUI_ACTIVE"): r["valu_busy"]=q["SQ_ACTIVE_INST_VALU"]/(1024.0*(g["GRBM_GUI_ACTIVE"]/8.0)/4.0)
    res[mode]=r
json.dump(res, open("gpurun_out/r06_event_team_pmc.json","w"), indent=1)
for m,r in res.items(): print(m, r)
PY
