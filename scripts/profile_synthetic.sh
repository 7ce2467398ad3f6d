#!/bin/bash
# usage (on the GPU box, through gpurun): scripts/profile_synthetic.sh
# The GLOBAL-memory roofline point (BASELINE.json configs[4]: 100k variables x 500k propagators, store 800 KB per workgroup):
# FETCH_SIZE / WRITE_SIZE counter passes of bench.py --workload synthetic, with the workgroups' stores inside the 256 MiB
# Infinity Cache (256 x 1024 threads: 205 MB) and far beyond it (256-thread workgroups: > 1 GB), sweeps and event fixpoint.
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $root
out=gpurun_out/prof_synth
rm -rf $out && mkdir -p $out
i=0
for cfg in "wac1 0" "event 0" "wac1 256" "event 256"; do
  set -- $cfg
  i=$((i+1))
  args="--workload synthetic --fixpoint $1 --threads $2 --steps 2 --warmup 1 --side-steps 0 --no-cpu-baseline"
  python3 bench.py $args > $out/plain$i.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $out/fetch$i -o p -- python3 bench.py $args > $out/fetch$i.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $out/write$i -o p -- python3 bench.py $args > $out/write$i.log 2>&1
done
python3 - <<PY
import glob, json, sqlite3
rows = []
cfgs = ["wac1 0", "event 0", "wac1 256", "event 256"]
for i, cfg in enumerate(cfgs, 1):
    def line(f):
        l = [x for x in open(f) if x.startswith("{")]
        return json.loads(l[-1]) if l else {}
    def counter(d, name):
        for db in glob.glob("$out/%s%d/**/p_results.db" % (d, i), recursive=True):
            con = sqlite3.connect(db)
            for tot, n in con.execute("select sum(value), count(*) from counters_collection where counter_name=? and kernel_name like '%solve_kernel%'", (name,)):
                return tot / max(1, n) if tot is not None else None
    b = line("$out/plain%d.log" % i)
    if not b: continue
    props = b["balance"]["propagations"]; ms = b["roofline"]["avg_launch_ms"]
    fetch, write = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
    bf = line("$out/fetch%d.log" % i)
    ms_f = bf.get("roofline", {}).get("avg_launch_ms", ms)
    props_f = bf.get("balance", {}).get("propagations", props)
    r = {"fixpoint": cfg.split()[0], "threads": b["config"]["workload"], "value": b["value"], "nodes_per_sec": b["nodes_per_sec"], "launch_ms": ms,
         "algorithmic_gbps": b["roofline"]["achieved"], "algorithmic_frac_of_hbm_peak": b["roofline"]["frac"],
         "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write}
    if fetch is not None:
        # profiles/r02_fetch_calibration.json: a random 8-byte gather is tallied as one 64-byte request (no correction);
        # the coalesced 16-byte record stream is tallied at half its bytes.  Records = 16 B x propagations.
        rec_bytes = 16.0 * props_f
        gathers_bytes = fetch * 1024.0 - rec_bytes / 2.0
        r["fabric_read_gbps"] = (gathers_bytes + rec_bytes) / (ms_f * 1e-3) / 1e9
        r["fabric_read_frac_of_hbm_peak"] = r["fabric_read_gbps"] / 8000.0
        r["fabric_read_bytes_per_propagation"] = (gathers_bytes + rec_bytes) / props_f
    rows.append(r)
json.dump({"note": "bench.py --workload synthetic (100 003 variables x 500 000 propagators, store in global memory); FETCH_SIZE interpreted with "
                   "profiles/r02_fetch_calibration.json; fabric reads include Infinity Cache hits", "rows": rows}, open("$out/r02_synthetic.json", "w"), indent=1)
print(json.dumps(rows, indent=1))
PY
