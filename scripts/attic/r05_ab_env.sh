#!/bin/bash
# Same-box A/B of an environment switch the engine reads at session creation (TB_NO_LEAN_MUL, TB_NO_COND_WAKE ...), production library:
#   scripts/r05_ab_env.sh VAR [workload] [nodes]
# twice each way without the profiler (nodes/s), then once each way under `rocprofv3 --pmc` (VALU / SALU wave-instructions per node).
var=$1; wl=${2:-wordpress7_500}; nodes=${3:-48000000}
root=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do
  for val in "" 1; do
    if [ -z "$val" ]; then unset $var; else export $var=1; fi
    python3 $root/scripts/valu_by_phase.py 0x0 $wl $nodes 2>&1 | grep -o "nodes=[0-9]* .*kernel_ns=[0-9]*" | awk -v tag="$var=${val:-unset}" '{split($1,a,"=");split($4,b,"=");printf "%s nodes/s %.4e\n", tag, a[2]/(b[2]*1e-9)}'
  done
done
for val in "" 1; do
  if [ -z "$val" ]; then unset $var; else export $var=1; fi
  out=$root/gpurun_out/abenv_${var}_${val:-unset}; rm -rf $out
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d $out -o p -- python3 $root/scripts/valu_by_phase.py 0x0 $wl 12000000 > $out.log 2>&1
  python3 - <<PY
import glob, sqlite3, re
log=open("$out.log").read()
m=re.search(r"nodes=(\d+) fails=(\d+) deductions=(\d+) kernel_ns=(\d+)", log)
n=int(m.group(1)); ns=int(m.group(4))
db=glob.glob("$out/**/*_results.db", recursive=True)[0]
con=sqlite3.connect(db)
c={k:v for k,v in con.execute("select counter_name, sum(value) from counters_collection where kernel_name like '%solve_kernel%' group by counter_name")}
print("$var=${val:-unset} (pmc)", "nodes/s %.3e" % (n/(ns*1e-9)), {k.replace("SQ_",""):round(v/n,1) for k,v in sorted(c.items())})
PY
done
