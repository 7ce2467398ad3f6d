#!/bin/bash
# usage (on the GPU box, through gpurun): scripts/calibrate_fetch.sh   -> gpurun_out/fetch_calib/ + profiles/r02_fetch_calibration.json
# Counter-only rocprofv3 passes (FETCH_SIZE alone) of scripts/micro/gather_calib.hip; see the header of that file.
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $root
out=gpurun_out/fetch_calib
rm -rf $out && mkdir -p $out
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $out/gather_calib scripts/micro/gather_calib.hip || exit 1
i=0
for cfg in "64 256 0" "4096 256 0" "64 0 1" "4096 0 1"; do
  i=$((i+1))
  rocprofv3 --pmc FETCH_SIZE -d $out/p$i -o p -- $out/gather_calib $cfg > $out/p$i.log 2>&1
done
python3 - <<PY
import glob, json, sqlite3
rows = []
for i in range(1, 5):
    line = [l for l in open("$out/p%d.log" % i) if l.startswith("{")]
    rec = json.loads(line[-1]) if line else {}
    fetch = None
    for db in glob.glob("$out/p%d/**/p_results.db" % i, recursive=True):
        d = sqlite3.connect(db)
        for name, val in d.execute("select counter_name, sum(value) from counters_collection where kernel_name like '%gather8%' or kernel_name like '%stream16%' group by counter_name"):
            if name == "FETCH_SIZE": fetch = val
    rec["FETCH_SIZE_raw"] = fetch
    if fetch is not None:
        rec["FETCH_SIZE_bytes_if_KiB"] = fetch * 1024.0
        rec["counter_over_useful_bytes"] = fetch * 1024.0 / rec["useful_bytes"]
        rec["counter_over_64B_lines"] = fetch * 1024.0 / (rec["accesses"] * 64.0) if rec["mode"].startswith("random") else None
    rows.append(rec)
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE on scripts/micro/gather_calib.hip (MI355X): the counter is reported in KiB; "
                   "counter_over_useful_bytes for the coalesced stream reproduces the guide's 0.5; for random 8-byte gathers each access moves "
                   "one memory-side request, counter_over_64B_lines says how that request is tallied", "rows": rows},
          open("profiles/r02_fetch_calibration.json", "w"), indent=1)
print(json.dumps(rows, indent=1))
PY
cp profiles/r02_fetch_calibration.json $out/
