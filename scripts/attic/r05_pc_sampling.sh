#!/bin/bash
# PC sampling of the headline search kernel (rocprofv3 beta feature; host-trap method: every wave's PC at a fixed time interval).  One bounded attempt:
# every step under its own timeout.  Needs a library with line tables: scripts/build_variant.sh pcs -gline-tables-only
#   scripts/r05_pc_sampling.sh [interval_us] [workload] [nodes] [fixpoint]
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $root
out=gpurun_out/pcs; rm -rf $out; mkdir -p $out
timeout 60 rocprofv3 -L > $out/avail.txt 2>&1; grep -i -A12 "pc.sampl" $out/avail.txt | head -60
export TURBO_HIP_LIB=$root/turbo_amd/lib/ab/pcs.so
timeout 120 python3 scripts/pcs_worker.py ${2:-example_wordpress7_500.fzn} ${3:-100000000} ${4:-2}   # the same launch without the profiler (rate reference)
for iv in ${1:-4096} 1000 65536; do   # (the accepted intervals are listed by `rocprofv3 -L`, above; the first that yields samples is kept)
  rm -rf $out/run
  timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval $iv --output-format csv -d $out/run -o p -- python3 scripts/pcs_worker.py ${2:-example_wordpress7_500.fzn} ${3:-100000000} ${4:-2} > $out/run.log 2>&1
  echo "pc sampling, interval $iv: rc=$?"; tail -5 $out/run.log
  [ -n "$(find $out/run -name '*pc_sampling*csv' -size +1k 2>/dev/null | head -1)" ] && break
done
find $out/run -type f | head -20
f=$(find $out/run -name "*pc_sampling*csv" | head -1)
[ -n "$f" ] && { wc -l $f; head -3 $f; python3 scripts/pcs_summary.py $f > $out/summary.txt 2>&1; head -80 $out/summary.txt; find $out/run -name "*.csv" -size +20M -delete; }
# second workload, only when the first run produced samples: the team kernel on the synthetic network (WAC1 sweeps)
if [ -n "$f" ] && [ -n "$PCS_SECOND" ]; then
  set -- $PCS_SECOND
  timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval $iv --output-format csv -d $out/run2 -o p -- python3 scripts/pcs_worker.py $1 $2 $3 > $out/run2.log 2>&1
  echo "pc sampling ($PCS_SECOND): rc=$?"; tail -3 $out/run2.log
  g=$(find $out/run2 -name "*pc_sampling*csv" | head -1)
  [ -n "$g" ] && { wc -l $g; python3 scripts/pcs_summary.py $g > $out/summary2.txt 2>&1; head -60 $out/summary2.txt; find $out/run2 -name "*.csv" -size +20M -delete; }
fi
