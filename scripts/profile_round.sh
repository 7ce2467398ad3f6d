#!/bin/bash
# usage (on the GPU box, through gpurun): scripts/profile_round.sh <tag>
# kernel trace of the default bench command, then FETCH_SIZE and WRITE_SIZE in their own counter-only passes
tag=${1:-r01}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $root
out=gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python3 bench.py > $out/bench_traced.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o f -- python3 bench.py --steps 2 --warmup 1 --event-steps 0 --no-cpu-baseline > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/write -o w -- python3 bench.py --steps 2 --warmup 1 --event-steps 0 --no-cpu-baseline > $out/write.log 2>&1
python3 scripts/summarize_profile.py $tag $(ls $out/trace/*/t_results.db $out/trace/t_results.db 2>/dev/null | head -1) $(ls $out/fetch/*/f_results.db $out/fetch/f_results.db 2>/dev/null | head -1) $(ls $out/write/*/w_results.db $out/write/w_results.db 2>/dev/null | head -1) --simplified
mkdir -p gpurun_out/profiles_$tag && cp profiles/${tag}_kernel_stats.txt profiles/${tag}_pmc.json profiles/pmc_traffic.json gpurun_out/profiles_$tag/
python3 bench.py > gpurun_out/profiles_$tag/bench_line.json 2> $out/bench.err
tail -c 3000 gpurun_out/profiles_$tag/bench_line.json
