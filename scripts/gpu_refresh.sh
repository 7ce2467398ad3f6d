#!/bin/bash
# refresh of the round's judged artefacts: trace + counters of the bench command, rates of every mode, bench line, budgets of a node
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash scripts/profile_r03.sh r03 > gpurun_out/refresh_profile.log 2>&1; echo "profile rc=$?"
timeout 900 python3 scripts/rates_table.py --out gpurun_out/r03_rates.json > gpurun_out/refresh_rates.log 2>&1; echo "rates rc=$?"
timeout 600 python3 bench.py > gpurun_out/r03_bench_final.json 2> gpurun_out/r03_bench_final.err; echo "bench rc=$?"; head -c 300 gpurun_out/r03_bench_final.json; echo
timeout 300 python3 scripts/time_budget.py wordpress7_500 24000000 gpurun_out/r03_time_budget.json > gpurun_out/refresh_time.log 2>&1; echo "time rc=$?"
if ls turbo_amd/lib/phases/phase_14.so > /dev/null 2>&1; then timeout 900 python3 scripts/phase_budget.py r03 wordpress7_500 24000000 > gpurun_out/refresh_phase.log 2>&1; echo "phase rc=$?"; fi
