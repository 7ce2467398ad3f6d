#!/bin/bash
# round-3 refresh, part 1: kernel trace + counters of the bench command, rates of every mode on every configuration, time budget
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash scripts/profile_r03.sh r03 > gpurun_out/final1_profile.log 2>&1; echo "profile rc=$?"; tail -3 gpurun_out/final1_profile.log
timeout 900 python3 scripts/rates_table.py --out gpurun_out/r03_rates.json > gpurun_out/final1_rates.log 2>&1; echo "rates rc=$?"; tail -30 gpurun_out/final1_rates.log
timeout 300 python3 scripts/time_budget.py wordpress7_500 24000000 gpurun_out/r03_time_budget.json > gpurun_out/final1_time.log 2>&1; echo "time rc=$?"; tail -60 gpurun_out/final1_time.log
