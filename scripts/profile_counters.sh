#!/bin/bash
# usage (on the GPU box, through gpurun): scripts/profile_counters.sh [tag]
# 1. kernel trace of the DEFAULT bench command (headline wordpress7_500 + side rows + other_workloads + cpu baseline + reference invocation);
# 2. counter-only passes (no tracing domains, one counter set per pass) of the headline per fixpoint: two SQ sets, instruction cache, GRBM clock,
#    FETCH_SIZE, WRITE_SIZE, L2 hit / miss -- `--reference-seconds 0 --other-steps 0`: no child process, one search kernel per pass;
# 3. the same memory-side passes for the synthetic 100k x 500k network (configs[4]: store in global memory), wac1 and event, stores inside
#    the Infinity Cache (256 x 1024 threads) and beyond it (256-thread workgroups);
# 4. scripts/summarize_counters.py -> profiles/<tag>_kernel_stats.txt, profiles/<tag>_counters.json.
# usage: scripts/profile_counters.sh [tag] [parts]   parts: any of t (trace) h (headline counters) o (accap_a3 / trains15 counters) s (synthetic); default "thos".
# Every profiled command runs under its own `timeout` (a profiler that hangs after a fault must not eat the GPU budget: the first final run of this round lost 55
# minutes that way), and the summary keeps the entries of profiles/<tag>_counters.json whose passes were not run again.
tag=${1:-r06}
parts=${2:-thos}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $root
out=gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
case $parts in *t*)
timeout 600 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python3 bench.py > $out/bench_traced.log 2> $out/bench_traced.err; echo "trace rc=$?";;
esac
pass() {  # pass <dir> <counters...> -- <bench args>
  d=$1; shift; c=""
  while [ "$1" != "--" ]; do c="$c $1"; shift; done; shift
  timeout 420 rocprofv3 --pmc $c -d $out/$d -o p -- python3 bench.py "$@" > $out/$d.log 2>&1 || echo "pass $d: rc=$?"
}
case $parts in *h*)
for fp in event wac1; do
  args="--steps 2 --warmup 1 --side-steps 0 --other-steps 0 --reference-seconds 0 --no-cpu-baseline --sharded-search 0 --fixpoint $fp"
  pass ${fp}_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU -- $args
  pass ${fp}_sq2 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAIT_INST_LDS -- $args
  pass ${fp}_icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -- $args
  pass ${fp}_grbm GRBM_GUI_ACTIVE -- $args
  pass ${fp}_fetch FETCH_SIZE -- $args
  pass ${fp}_write WRITE_SIZE -- $args
  pass ${fp}_tcc TCC_HIT_sum TCC_MISS_sum -- $args
done;;
esac
case $parts in *o*)
for w in accap_a3 trains15; do  # the other LDS-resident BASELINE configurations, event fixpoint
  args="--workload $w --steps 2 --warmup 1 --side-steps 0 --other-steps 0 --reference-seconds 0 --no-cpu-baseline --sharded-search 0 --fixpoint event"
  pass ${w}_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU -- $args
  pass ${w}_sq2 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAIT_INST_LDS -- $args
  pass ${w}_grbm GRBM_GUI_ACTIVE -- $args
  pass ${w}_fetch FETCH_SIZE -- $args
  pass ${w}_write WRITE_SIZE -- $args
done;;
esac
case $parts in *s*)
# the synthetic 100k x 500k network (configs[4]): WAC1 and AC1 sweeps in workgroup teams, the event fixpoint on the hot tier -- memory side AND issue side (r06: VERDICT r05 item 4)
for cfg in "1 wac1" "2 event" "5 ac1"; do
  set -- $cfg
  i=$1
  args="--workload synthetic --fixpoint $2 --steps 2 --warmup 1 --side-steps 0 --other-steps 0 --reference-seconds 0 --no-cpu-baseline --sharded-search 0"
  python3 bench.py $args > $out/syn${i}_plain.log 2>&1
  pass syn${i}_fetch FETCH_SIZE -- $args
  pass syn${i}_write WRITE_SIZE -- $args
  pass syn${i}_tcc TCC_HIT_sum TCC_MISS_sum -- $args
  pass syn${i}_ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -- $args
  pass syn${i}_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU -- $args
  pass syn${i}_sq2 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAIT_INST_LDS -- $args
  pass syn${i}_grbm GRBM_GUI_ACTIVE -- $args
done;;
esac
unset TB_TEAM
python3 scripts/summarize_counters.py $tag $out
mkdir -p gpurun_out/profiles_$tag && cp profiles/${tag}_kernel_stats.txt profiles/${tag}_counters.json gpurun_out/profiles_$tag/
