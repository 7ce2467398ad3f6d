#!/usr/bin/env python3
"""Turn the rocprofv3 rocpd databases of a profiled bench run into small committed summaries.

usage: summarize_profile.py <round tag> <kernel-trace db> [<FETCH_SIZE db> <WRITE_SIZE db>] [--workload W --cutnodes C --fixpoint F]
Writes profiles/<tag>_kernel_stats.txt, profiles/<tag>_pmc.json and profiles/pmc_traffic.json.
HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE come from separate --pmc
passes, are in KiB, and on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads (x2).
"""
import argparse, json, os, sqlite3, sys

ap = argparse.ArgumentParser()
ap.add_argument("tag"); ap.add_argument("trace_db"); ap.add_argument("fetch_db", nargs="?"); ap.add_argument("write_db", nargs="?")
ap.add_argument("--workload", default="wordpress7_500"); ap.add_argument("--cutnodes", type=int, default=3000); ap.add_argument("--fixpoint", default="wac1")
ap.add_argument("--kernel", default="solve_kernel"); ap.add_argument("--simplified", action="store_true")
a = ap.parse_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(root, "profiles"); os.makedirs(out_dir, exist_ok=True)

db = sqlite3.connect(a.trace_db)
rows = list(db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
with open(os.path.join(out_dir, f"{a.tag}_kernel_stats.txt"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py (workload {a.workload}, cutnodes {a.cutnodes}, {a.fixpoint}); durations in us\n")
    f.write(f"{'kernel':70s} {'calls':>6s} {'total_us':>14s} {'avg_us':>14s} {'pct':>8s}\n")
    for n, c, t, avg, p in rows:
        f.write(f"{n[:70]:70s} {c:6d} {t:14.1f} {avg:14.1f} {p:8.3f}\n")
    disp = list(db.execute("select name, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count from kernels where name like ? limit 1", (f"%{a.kernel}%",)))
    if disp:
        f.write(f"\n# dispatch of the dominant kernel: grid={disp[0][1]} workgroup={disp[0][2]} lds_block_size={disp[0][3]} scratch={disp[0][4]} arch_vgpr={disp[0][5]} accum_vgpr={disp[0][6]} sgpr={disp[0][7]}\n")
print(open(os.path.join(out_dir, f"{a.tag}_kernel_stats.txt")).read())

def per_launch(dbfile, counter):
    d = sqlite3.connect(dbfile)
    vals = [r[0] for r in d.execute("select value from counters_collection where counter_name=? and kernel_name like ? order by start", (counter, f"%{a.kernel}%"))]
    return vals

if a.fetch_db and a.write_db:
    fetch = per_launch(a.fetch_db, "FETCH_SIZE"); write = per_launch(a.write_db, "WRITE_SIZE")
    f_kib = sum(fetch) / max(1, len(fetch)); w_kib = sum(write) / max(1, len(write))
    hbm = (2.0 * f_kib + w_kib) * 1024.0
    rec = {"tag": a.tag, "workload": a.workload, "cutnodes": a.cutnodes, "fixpoint": a.fixpoint, "simplified": a.simplified, "kernel": a.kernel,
           "launches_profiled": len(fetch), "FETCH_SIZE_KiB_per_launch": f_kib, "WRITE_SIZE_KiB_per_launch": w_kib,
           "correction": "hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x)",
           "hbm_bytes_per_launch": hbm}
    json.dump(rec, open(os.path.join(out_dir, f"{a.tag}_pmc.json"), "w"), indent=1)
    json.dump(rec, open(os.path.join(out_dir, "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(rec, indent=1))
