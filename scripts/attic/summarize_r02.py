#!/usr/bin/env python3
"""profiles/<tag>_kernel_stats.txt and profiles/<tag>_counters.json from the rocprofv3 databases written by scripts/profile_r02.sh.

Per fixpoint mode of the default bench workload, per launch of tb::solve_kernel (averages over the profiled launches):
  launch_ms                     average duration in the counter passes (rocprofv3 dispatch timestamps)
  valu_busy / salu_busy / lds_busy   SQ_ACTIVE_INST_{VALU,SCA,LDS} / (1024 SIMDs x (GRBM_GUI_ACTIVE / 8 XCDs) / 4): the counters are in quad-cycles
  wait_any_share, wait_inst_any_share   SQ_WAIT_ANY, SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (waves parked at s_waitcnt / s_barrier; issue stalls)
  valu_per_64_propagations ...   SQ_INSTS_* / (num_deductions / 64), num_deductions from the bench line of the same pass
  hbm_bytes_per_launch          (2 x FETCH_SIZE + WRITE_SIZE) x 1024: MI355X_MICROARCH.md's gfx950 correction for wide coalesced reads -- the
                                HBM traffic of these kernels is the record stream and the snapshot copies, both 16 B per lane
"""
import glob, json, os, re, sqlite3, sys

tag, out = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
os.makedirs(prof, exist_ok=True)


def db_of(d):
    c = glob.glob(os.path.join(out, d, "**", "*_results.db"), recursive=True)
    return c[0] if c else None


def counters(d, kernel="solve_kernel"):
    path = db_of(d)
    if not path:
        return {}
    con = sqlite3.connect(path)
    res = {}
    for name, total, n in con.execute("select counter_name, sum(value), count(*) from counters_collection where kernel_name like ? group by counter_name", (f"%{kernel}%",)):
        res[name] = total / max(1, n)
    try:
        durs = [r[0] for r in con.execute("select (end - start) from kernels where name like ?", (f"%{kernel}%",))]
        if durs:
            res["_launch_ms"] = sum(durs) / len(durs) / 1e6
    except Exception:
        pass
    return res


def bench_line(log):
    try:
        lines = [l for l in open(os.path.join(out, log)) if l.startswith("{")]
        return json.loads(lines[-1]) if lines else {}
    except Exception:
        return {}


trace = db_of("trace")
if trace:
    con = sqlite3.connect(trace)
    rows = list(con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    with open(os.path.join(prof, f"{tag}_kernel_stats.txt"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py (default command: wordpress7_500 simplified, event fixpoint, wac1 beside it); durations in us\n")
        f.write(f"{'kernel':78s} {'calls':>6s} {'total_us':>14s} {'avg_us':>14s} {'pct':>8s}\n")
        for n, c, t, avg, p in rows:
            f.write(f"{n[:78]:78s} {c:6d} {t:14.1f} {avg:14.1f} {p:8.3f}\n")
        for n, gx, wx, lds, scr, vg, av, sg in con.execute("select name, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count from kernels where name like '%solve_kernel%' group by name"):
            f.write(f"\n# dispatch {n[:60]}: grid={gx} workgroup={wx} lds_block_size={lds} scratch={scr} arch_vgpr={vg} accum_vgpr={av} sgpr={sg}")
        f.write("\n")
        line = bench_line("bench_traced.log")
        if line:
            f.write(f"# bench line of the traced run: value={line.get('value'):.4e} propagations/s, nodes_per_sec={line.get('nodes_per_sec'):.4e}, roofline.avg_launch_ms={line['roofline']['avg_launch_ms']:.3f}\n")
    print(open(os.path.join(prof, f"{tag}_kernel_stats.txt")).read())

rec = {"note": __doc__.strip().split("\n\n")[1] if False else "rocprofv3 --pmc passes of `python3 bench.py --steps 2 --warmup 1 --side-steps 0 --no-cpu-baseline --fixpoint <mode>` "
               "(scripts/profile_r02.sh); one counter set per pass, no tracing domains; per-launch averages of tb::solve_kernel"}
for fp in ("event", "wac1"):
    sq1, sq2, grbm, fetch, write = (counters(f"{fp}_{k}") for k in ("sq1", "sq2", "grbm", "fetch", "write"))
    line = bench_line(f"{fp}_sq1.log")
    if not sq1 or not grbm:
        continue
    gui = grbm.get("GRBM_GUI_ACTIVE", 0.0) / 8.0  # the counter is summed over the 8 XCDs
    cap = 1024.0 * gui / 4.0  # quad-cycles available to the 1024 SIMDs during one launch
    launch_ms = sq1.get("_launch_ms")
    props = line.get("balance", {}).get("propagations")
    wave_props = props / 64.0 if props else None
    r = {"launch_ms": launch_ms, "gfx_clock_ghz": gui / (launch_ms * 1e-3) / 1e9 if launch_ms else None,
         "valu_busy": sq1["SQ_ACTIVE_INST_VALU"] / cap if cap else None, "salu_busy": sq2.get("SQ_ACTIVE_INST_SCA", 0) / cap if cap else None,
         "lds_busy": sq2.get("SQ_ACTIVE_INST_LDS", 0) / cap if cap else None,
         "wait_any_share": sq1["SQ_WAIT_ANY"] / sq1["SQ_WAVE_CYCLES"], "wait_inst_any_share": sq1["SQ_WAIT_INST_ANY"] / sq1["SQ_WAVE_CYCLES"],
         "lds_bank_conflict_share": sq2.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, sq2.get("SQ_LDS_IDX_ACTIVE", 1)),
         "nodes_per_launch": line.get("balance", {}).get("nodes"), "propagations_per_launch": props,
         "bench_value": line.get("value"), "bench_nodes_per_sec": line.get("nodes_per_sec"),
         "counters": {k: v for d in (sq1, sq2, grbm) for k, v in d.items() if not k.startswith("_")}}
    if wave_props:
        r["valu_per_64_propagations"] = sq1["SQ_INSTS_VALU"] / wave_props
        r["salu_per_64_propagations"] = sq1["SQ_INSTS_SALU"] / wave_props
        r["valu_per_node"] = sq1["SQ_INSTS_VALU"] / r["nodes_per_launch"]
        r["salu_per_node"] = sq1["SQ_INSTS_SALU"] / r["nodes_per_launch"]
    if fetch.get("FETCH_SIZE") is not None and write.get("WRITE_SIZE") is not None:
        hbm = (2.0 * fetch["FETCH_SIZE"] + write["WRITE_SIZE"]) * 1024.0
        ms = fetch.get("_launch_ms") or launch_ms
        r.update({"FETCH_SIZE_KiB": fetch["FETCH_SIZE"], "WRITE_SIZE_KiB": write["WRITE_SIZE"], "hbm_bytes_per_launch": hbm,
                  "hbm_gbps": hbm / (ms * 1e-3) / 1e9 if ms else None, "hbm_frac_of_peak": hbm / (ms * 1e-3) / 1e9 / 8000.0 if ms else None,
                  "hbm_note": "memory-side requests of the L2 (Infinity Cache hits included): record stream, snapshot copies, and the register spills of "
                              "the event kernel (scratch); WRITE_SIZE is not calibrated"})
    rec[f"wordpress7_500/{fp}"] = r
json.dump(rec, open(os.path.join(prof, f"{tag}_counters.json"), "w"), indent=1)
print(json.dumps({k: {a: b for a, b in v.items() if a != "counters"} if isinstance(v, dict) else v for k, v in rec.items()}, indent=1))
