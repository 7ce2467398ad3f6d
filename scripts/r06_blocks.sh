#!/bin/bash
# Instruction budget of the event kernels from basic-block execution counts (scripts/instr_blocks.py): per workload, the hardware counters of the production library
# and the block counts of the instrumented library on the SAME budgeted search, then the table.  scripts/r06_blocks.sh [nodes] [workloads...]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
nodes=${1:-12000000}; shift
wl=${@:-wordpress7_500 trains15 accap_a3}
turbo_amd/bin/satomic_test > gpurun_out/r06_satomic.log 2>&1; cat gpurun_out/r06_satomic.log
for w in $wl; do
  extra=""; wf=$w
  case $w in *_proof) wf=${w%_proof}; extra="proof:500:21";; esac   # (wordpress7_500_proof: the timing-independent whole search that refutes objective <= 500)
  out=$GRAFT_REPO_ROOT/gpurun_out/r06_pmc_$w; rm -rf $out
  (cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $out -o p -- python3 $GRAFT_REPO_ROOT/scripts/valu_by_phase.py 0x0 $wf $nodes $extra > $out.log 2>&1)
  python3 - <<PY > gpurun_out/r06_pmc_$w.json
import glob, sqlite3, re, json
log=open("$out.log").read()
m=re.search(r"nodes=(\d+) fails=(\d+) deductions=(\d+) kernel_ns=(\d+)", log)
n=int(m.group(1)); ns=int(m.group(4))
db=glob.glob("$out/**/*_results.db", recursive=True)[0]
con=sqlite3.connect(db)
c={k:v for k,v in con.execute("select counter_name, sum(value) from counters_collection where kernel_name like '%solve_kernel%' group by counter_name")}
print(json.dumps({"workload": "$w", "nodes": n, "kernel_ns": ns, "nodes_per_sec": n/(ns*1e-9), "per_node": {k: v/n for k,v in sorted(c.items())}}))
PY
  cat gpurun_out/r06_pmc_$w.json
  TURBO_HIP_LIB=$GRAFT_REPO_ROOT/turbo_amd/lib/libturbo_hip_blocks.so TB_BLOCK_COUNTS=$GRAFT_REPO_ROOT/gpurun_out/r06_blocks_$w.bin timeout 600 python3 scripts/valu_by_phase.py 0x0 $wf $nodes $extra > gpurun_out/r06_blocks_$w.log 2>&1
  tail -2 gpurun_out/r06_blocks_$w.log
  python3 - <<PY
import json, subprocess, re
p=json.load(open("gpurun_out/r06_pmc_$w.json"))
log=open("gpurun_out/r06_blocks_$w.log").read()
n=int(re.search(r"nodes=(\d+)", log).group(1))
kern={"wordpress7_500": "ILi1ELi128ELb1ELi1E", "trains15": "ILi1ELi128ELb1ELi4E", "accap_a3": "ILi1ELi128ELb1ELi0E"}["$wf"]
out=subprocess.run(["python3","scripts/instr_blocks.py","table","turbo_amd/lib/blocks_segments.json","gpurun_out/r06_blocks_$w.bin",kern,str(n),str(p["per_node"]["SQ_INSTS_VALU"]),str(p["per_node"]["SQ_INSTS_SALU"])],capture_output=True,text=True)
open("gpurun_out/r06_region_budget_$w.json","w").write(out.stdout)
print(out.stderr[-2000:])
d=json.loads(out.stdout)
print("$w", d["sum"], d.get("measured"))
for r in d["regions"][:14]: print("  ", r["region"], r["what"][:50], {k: r[k] for k in ("valu","lane","salu","branch","misc","lds") if k in r})
PY
done
