#!/bin/bash
# usage (on the GPU box, through gpurun): scripts/profile_r02.sh [tag]
# 1. kernel trace of the default bench command (headline: event fixpoint; beside it: wac1);
# 2. counter-only passes (no tracing domains) per fixpoint: two SQ sets, GRBM clock, FETCH_SIZE, WRITE_SIZE;
# 3. scripts/summarize_r02.py -> profiles/<tag>_kernel_stats.txt, profiles/<tag>_counters.json.
tag=${1:-r02}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $root
out=gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python3 bench.py > $out/bench_traced.log 2>&1
for fp in event wac1; do
  args="--steps 2 --warmup 1 --side-steps 0 --no-cpu-baseline --fixpoint $fp"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU -d $out/${fp}_sq1 -o p -- python3 bench.py $args > $out/${fp}_sq1.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAIT_INST_LDS -d $out/${fp}_sq2 -o p -- python3 bench.py $args > $out/${fp}_sq2.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE -d $out/${fp}_grbm -o p -- python3 bench.py $args > $out/${fp}_grbm.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $out/${fp}_fetch -o p -- python3 bench.py $args > $out/${fp}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $out/${fp}_write -o p -- python3 bench.py $args > $out/${fp}_write.log 2>&1
done
python3 scripts/summarize_r02.py $tag $out
mkdir -p gpurun_out/profiles_$tag && cp profiles/${tag}_kernel_stats.txt profiles/${tag}_counters.json gpurun_out/profiles_$tag/
