cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_headline_trees.py tests/test_gpu_selfcheck.py tests/test_gpu_fullgrid_paths.py tests/test_gpu_parity.py -x -q -k "not soak" > gpurun_out/r06e_t.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r06e_t.log
timeout 1500 bash scripts/r06_blocks.sh 12000000 > gpurun_out/r06_blocks.log 2>&1; echo "blocks rc=$?"; grep "^wordpress\|^trains\|^accap" gpurun_out/r06_blocks.log | cut -c1-500
timeout 300 python3 bench.py --steps 3 --warmup 1 --side-steps 0 --other-steps 0 --no-cpu-baseline --reference-seconds 0 --sharded-search 0 > gpurun_out/r06_bench1.json 2> gpurun_out/r06_bench1.err; echo "bench rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/r06_bench1.json')); print('nodes/s %.4e props/s %.4e ms %.1f' % (d['nodes_per_sec'], d['value'], d['ms_per_step']))"
for w in trains15 accap_a3; do timeout 200 python3 scripts/quick_rate.py $w nodes=48000000 fixpoint=2 2>&1 | tail -1; done
