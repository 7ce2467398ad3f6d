#!/bin/bash
# same-box A/B of engine builds: scripts/r04_ab_libs.sh tag lib1.so lib2.so ...   (two interleaved rounds of the headline bench without side rows)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=$1; shift
for round in 1 2; do
  for lib in "$@"; do
    name=$(basename $lib .so)
    TURBO_HIP_LIB=$lib timeout 300 python3 bench.py --steps 3 --warmup 1 --side-steps 0 --other-steps 0 --no-cpu-baseline --reference-seconds 0 > gpurun_out/${tag}_${name}_$round.json 2> gpurun_out/${tag}_${name}_$round.err
    python3 - <<PY
import json
try:
    d=json.load(open("gpurun_out/${tag}_${name}_$round.json"))
    print("$name round $round: nodes/s %.4e  props/s %.4e  evals/node %.0f" % (d["nodes_per_sec"], d["value"], d["value"]/d["nodes_per_sec"]))
except Exception as e:
    print("$name round $round: FAILED", e); print(open("gpurun_out/${tag}_${name}_$round.err").read()[-600:])
PY
  done
done
