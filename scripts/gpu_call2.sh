#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; tag=${1:-c2}
timeout 1200 python3 -m pytest tests -m gpu -x -q > gpurun_out/${tag}_t.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/${tag}_t.log
timeout 300 python3 bench.py --steps 3 --warmup 1 --side-steps 0 --no-cpu-baseline --reference-seconds 0 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/${tag}_bench.json')); print('nodes/s %.4e  props/s %.4e  ms/step %.1f' % (d['nodes_per_sec'], d['value'], d['ms_per_step']))"
timeout 600 python3 scripts/phase_budget.py ${tag} wordpress7_500 12000000 > gpurun_out/${tag}_phase.log 2>&1; echo "phase rc=$?"
python3 - <<PY
import json
d=json.load(open('gpurun_out/${tag}_phase_budget_wordpress7_500.json'))
for k,v in d["per_node_by_phase"].items(): print(k, {a:round(b,3) for a,b in v.items()})
b=d["runs"]["base"]; print({k:round(v,1) for k,v in b.items() if isinstance(v,float)})
PY
