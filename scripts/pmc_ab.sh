#!/bin/bash
# same-box instruction counters of engine builds: scripts/pmc_ab.sh lib1.so lib2.so ... (production-style builds, knob 0)
cd /tmp; export TMPDIR=/tmp
for lib in "$@"; do
  out=$GRAFT_REPO_ROOT/gpurun_out/pmcab_$(basename ${lib%.so}); rm -rf $out
  export TURBO_HIP_LIB=$GRAFT_REPO_ROOT/turbo_amd/lib/$lib
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d $out -o p -- python3 $GRAFT_REPO_ROOT/scripts/valu_by_phase.py 0x0 ${PMC_WORKLOAD:-wordpress7_500} ${PMC_NODES:-12000000} > $out.log 2>&1
  python3 - <<PY
import glob, sqlite3, re
log=open("$out.log").read()
m=re.search(r"nodes=(\d+) fails=(\d+) deductions=(\d+) kernel_ns=(\d+)", log)
n=int(m.group(1)); ns=int(m.group(4))
db=glob.glob("$out/**/*_results.db", recursive=True)[0]
con=sqlite3.connect(db)
c={k:v for k,v in con.execute("select counter_name, sum(value) from counters_collection where kernel_name like '%solve_kernel%' group by counter_name")}
print("$lib", "nodes/s %.3e" % (n/(ns*1e-9)), {k.replace("SQ_",""):round(v/n,1) for k,v in sorted(c.items())})
PY
done
