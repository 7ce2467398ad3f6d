cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_team.py tests/test_gpu_fullsize_global.py -x -q -s > gpurun_out/r06a_t.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r06a_t.log
timeout 700 bash scripts/r06_solve_timeline.sh > gpurun_out/r06_tl.log 2>&1; echo "timeline rc=$?"; tail -40 gpurun_out/r06_tl.log
timeout 400 python3 bench.py --steps 5 --warmup 1 > gpurun_out/r06_bench0.json 2> gpurun_out/r06_bench0.err; echo "bench rc=$?"; tail -3 gpurun_out/r06_bench0.err
python3 - <<PY
import json
d=json.load(open("gpurun_out/r06_bench0.json"))
print("nodes/s %.4e  props/s %.4e  ms/step %.1f" % (d["nodes_per_sec"], d["value"], d["ms_per_step"]))
print(json.dumps(d.get("sharded_search"))[:1500])
for o in d.get("other_workloads", []): print(o["workload"][:30], o["fixpoint"], "%.3e" % o["nodes_per_sec"], o["roofline"]["bound"], "%.3f" % o["roofline"]["frac"])
PY
