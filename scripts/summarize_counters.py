#!/usr/bin/env python3
"""profiles/<tag>_kernel_stats.txt and profiles/<tag>_counters.json from the rocprofv3 databases written by scripts/profile_counters.sh.

Per fixpoint mode of the headline workload (and per configuration of the synthetic 100k x 500k one), per launch of tb::solve_kernel (averages
over the profiled launches of the pass):
  launch_ms                     average duration in the counter passes (rocprofv3 dispatch timestamps)
  valu_busy / salu_busy / lds_busy   SQ_ACTIVE_INST_{VALU,SCA,LDS} / (1024 SIMDs x (GRBM_GUI_ACTIVE / 8 XCDs) / 4): the counters are in quad-cycles
  wait_any_share, wait_inst_any_share   SQ_WAIT_ANY, SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (waves parked at s_waitcnt / s_barrier; issue stalls)
  valu_per_node ...             SQ_INSTS_* / nodes of the launch (bench line of the same pass)
  icache_hit_rate               SQC_ICACHE_HITS / SQC_ICACHE_REQ
  hbm_bytes_per_launch          (2 x FETCH_SIZE + WRITE_SIZE) x 1024 for the LDS-resident workloads (16-byte-per-lane streams: the guide's gfx950 correction);
                                for the synthetic network `fabric_read_bytes_per_propagation` = FETCH_SIZE x 1024 / propagations, uncorrected: its reads are
                                random 8-byte gathers, which FETCH_SIZE tallies as one 64-byte request each (profiles/r02_fetch_calibration.json)
  tcc_hit_rate                  TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
"""
import glob, json, os, sqlite3, sys

tag, out = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
os.makedirs(prof, exist_ok=True)


def db_of(d, kernel="solve_kernel"):
    """The database of a pass: the one holding the most dispatches of the search kernel (a pass may leave more than one)."""
    best, rows = None, -1
    for c in glob.glob(os.path.join(out, d, "**", "*_results.db"), recursive=True):
        try:
            n = sqlite3.connect(c).execute("select count(*) from kernels where name like ?", (f"%{kernel}%",)).fetchone()[0]
        except Exception:
            n = 0
        if n > rows:
            best, rows = c, n
    return best


def counters(d, kernel="solve_kernel"):
    path = db_of(d, kernel)
    if not path:
        return {}
    con = sqlite3.connect(path)
    res = {}
    try:
        for name, total, n in con.execute("select counter_name, sum(value), count(*) from counters_collection where kernel_name like ? group by counter_name", (f"%{kernel}%",)):
            res[name] = total / max(1, n)
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
        f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py (default command: wordpress7_500 simplified, event fixpoint; wac1 / ac1 side rows; other_workloads: accap_a3, trains15, synthetic); durations in us\n")
        f.write(f"{'kernel':78s} {'calls':>6s} {'total_us':>14s} {'avg_us':>14s} {'pct':>8s}\n")
        for n, c, t, avg, p in rows:
            f.write(f"{n[:78]:78s} {c:6d} {t:14.1f} {avg:14.1f} {p:8.3f}\n")
        f.write("\n# dispatch geometry of the search kernels (one line per distinct launch shape)\n")
        for n, gx, wx, lds, scr, vg, av, sg, cnt, avg in con.execute("select name, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count, count(*), avg(end - start) from kernels "
                                                                     "where name like '%solve_kernel%' group by name, grid_x, workgroup_x, lds_size"):
            f.write(f"# {n[:58]}: grid={gx} workgroup={wx} lds_block_size={lds} scratch={scr} arch_vgpr={vg} accum_vgpr={av} sgpr={sg} launches={cnt} avg_ms={avg / 1e6:.3f}\n")
        line = bench_line("bench_traced.log")
        # The headline kernel is launched by two parts of the command: the timed loop (warm-up + steps: the node budget of a step) and the `sharded_search` record (a whole
        # proof under a fixed bound, a different amount of work per launch) -- the table above averages over both; here they are apart, in launch order.
        if line:
            head = next(iter(r[0] for r in con.execute("select name from kernels where name like '%solve_kernel%' order by start limit 1")), None)  # (the loop runs first)
            durs = [r[0] / 1e6 for r in con.execute("select (end - start) from kernels where name = ? order by start", (head,))]
            k = int(line.get("warmup", 0)) + int(line.get("steps", 0))
            loop, rest = durs[:k], durs[k:]
            f.write(f"# launches of the headline kernel in order, ms: {' '.join(f'{d:.1f}' for d in durs)}\n")
            if loop:
                timed = loop[int(line.get('warmup', 0)):]
                f.write(f"#   timed loop ({line.get('warmup')} warm-up + {line.get('steps')} steps): average of the {len(timed)} timed launches {sum(timed) / max(1, len(timed)):.3f} ms"
                        f" -- against roofline.avg_launch_ms of the same run (HIP events in bench.py) {line['roofline']['avg_launch_ms']:.3f} ms\n")
            if rest:
                f.write(f"#   sharded_search record (warm-up + runs of the proof under a fixed bound): {len(rest)} launches, average {sum(rest) / len(rest):.3f} ms"
                        f" (the record's seconds: {line.get('sharded_search', {}).get('seconds')})\n")
            f.write(f"# bench line of the traced run: value={line.get('value'):.4e} propagations/s, nodes_per_sec={line.get('nodes_per_sec'):.4e}, roofline.avg_launch_ms={line['roofline']['avg_launch_ms']:.3f}\n")
            json.dump(line, open(os.path.join(prof, f"{tag}_bench_line_traced.json"), "w"), indent=1)
    print(open(os.path.join(prof, f"{tag}_kernel_stats.txt")).read())

try:  # entries of an earlier run of this tag stay when their passes were not run again (scripts/profile_counters.sh: parts)
    earlier = json.load(open(os.path.join(prof, f"{tag}_counters.json")))
except Exception:
    earlier = {}
rec = {"note": "rocprofv3 --pmc passes of `python3 bench.py --steps 2 --warmup 1 --side-steps 0 --other-steps 0 --reference-seconds 0 --no-cpu-baseline --fixpoint <mode>` "
               "(scripts/profile_counters.sh); one counter set per pass, no tracing domains; per-launch averages of tb::solve_kernel"}
for fp, key in (("event", "wordpress7_500/event"), ("wac1", "wordpress7_500/wac1"), ("accap_a3", "accap_a3/event"), ("trains15", "trains15/event")):
    sq1, sq2, ic, grbm, fetch, write, tcc = (counters(f"{fp}_{k}") for k in ("sq1", "sq2", "icache", "grbm", "fetch", "write", "tcc"))
    line = bench_line(f"{fp}_sq1.log")
    if not sq1 or not grbm:
        continue
    gui = grbm.get("GRBM_GUI_ACTIVE", 0.0) / 8.0  # the counter is summed over the 8 XCDs
    cap = 1024.0 * gui / 4.0  # quad-cycles available to the 1024 SIMDs during one launch
    launch_ms = sq1.get("_launch_ms")
    props = line.get("balance", {}).get("propagations")
    nodes = line.get("balance", {}).get("nodes")
    r = {"launch_ms": launch_ms, "gfx_clock_ghz": gui / (launch_ms * 1e-3) / 1e9 if launch_ms else None,
         "valu_busy": sq1["SQ_ACTIVE_INST_VALU"] / cap if cap else None, "salu_busy": sq2.get("SQ_ACTIVE_INST_SCA", 0) / cap if cap else None,
         "lds_busy": sq2.get("SQ_ACTIVE_INST_LDS", 0) / cap if cap else None,
         "wait_any_share": sq1["SQ_WAIT_ANY"] / sq1["SQ_WAVE_CYCLES"], "wait_inst_any_share": sq1["SQ_WAIT_INST_ANY"] / sq1["SQ_WAVE_CYCLES"],
         "lds_bank_conflict_share": sq2.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, sq2.get("SQ_LDS_IDX_ACTIVE", 1)),
         "nodes_per_launch": nodes, "propagations_per_launch": props,
         "bench_value": line.get("value"), "bench_nodes_per_sec": line.get("nodes_per_sec"),
         "evaluations_per_node": line.get("evaluations_per_node"),
         "counters": {k: v for d in (sq1, sq2, ic, grbm, tcc) for k, v in d.items() if not k.startswith("_")}}
    if props and nodes:
        r["valu_per_64_propagations"] = sq1["SQ_INSTS_VALU"] / (props / 64.0)
        r["salu_per_64_propagations"] = sq1["SQ_INSTS_SALU"] / (props / 64.0)
        r["valu_per_node"] = sq1["SQ_INSTS_VALU"] / nodes
        r["salu_per_node"] = sq1["SQ_INSTS_SALU"] / nodes
    if ic.get("SQC_ICACHE_REQ"):
        r["icache_hit_rate"] = ic.get("SQC_ICACHE_HITS", 0.0) / ic["SQC_ICACHE_REQ"]
        r["icache_misses_per_node"] = (ic.get("SQC_ICACHE_MISSES", 0.0) / nodes) if nodes else None
    if tcc.get("TCC_HIT_sum") is not None:
        r["tcc_hit_rate"] = tcc["TCC_HIT_sum"] / max(1.0, tcc["TCC_HIT_sum"] + tcc.get("TCC_MISS_sum", 0.0))
    if fetch.get("FETCH_SIZE") is not None and write.get("WRITE_SIZE") is not None:
        hbm = (2.0 * fetch["FETCH_SIZE"] + write["WRITE_SIZE"]) * 1024.0
        ms = fetch.get("_launch_ms") or launch_ms
        r.update({"FETCH_SIZE_KiB": fetch["FETCH_SIZE"], "WRITE_SIZE_KiB": write["WRITE_SIZE"], "hbm_bytes_per_launch": hbm,
                  "hbm_bytes_per_node": hbm / nodes if nodes else None,
                  "hbm_gbps": hbm / (ms * 1e-3) / 1e9 if ms else None, "hbm_frac_of_peak": hbm / (ms * 1e-3) / 1e9 / 8000.0 if ms else None,
                  "hbm_note": "memory-side requests of the L2 (Infinity Cache hits included): record stream, snapshot copies, best-store copies and what is left of the "
                              "register spills (scratch); WRITE_SIZE is not calibrated"})
    rec[key] = r

team_round = tag >= "r05"  # r05: the sweeps are planned in workgroup teams; pass 4 is the hot tier (TB_TEAM=0), pass 5 the plain sweeps
for i, (fp, what) in enumerate((("wac1", "256 workgroups x 1024 threads in workgroup teams (four per XCD, 32 shared stores of 800 KB; solve_kernel_team)" if team_round else
                                         "256 workgroups x 1024 threads, hot tier (19 456 most-read intervals in LDS): the stores (205 MB) inside the Infinity Cache"),
                                ("event", "256 workgroups x 1024 threads, hot tier, event fixpoint"),
                                ("wac1", "256-thread workgroups (no hot tier): > 1 GB of stores, beyond the Infinity Cache"),
                                ("wac1", "256 workgroups x 1024 threads, hot tier (TB_TEAM=0: r04's plan; 205 MB of stores inside the Infinity Cache)" if team_round else
                                         "256 workgroups x 1024 threads WITHOUT the hot tier (TB_NO_HOT_TIER: r03's configuration)"),
                                ("ac1", "workgroup teams, plain sweeps (operands gathered one slice ahead)")), 1):
    b = bench_line(f"syn{i}_plain.log")
    if not b:
        continue
    fetch, write, tcc, ea = (counters(f"syn{i}_{k}") for k in ("fetch", "write", "tcc", "ea"))
    bf = bench_line(f"syn{i}_fetch.log")
    props_f = bf.get("balance", {}).get("propagations") or b["balance"]["propagations"]
    r = {"what": what, "workload": b["config"]["workload"], "value": b["value"], "nodes_per_sec": b["nodes_per_sec"], "launch_ms": b["roofline"]["avg_launch_ms"],
         "algorithmic_gbps": b["roofline"]["achieved"], "algorithmic_frac_of_hbm_peak": b["roofline"]["frac"]}
    if fetch.get("FETCH_SIZE") is not None:
        ms = fetch.get("_launch_ms") or b["roofline"]["avg_launch_ms"]
        r.update({"FETCH_SIZE_KiB": fetch["FETCH_SIZE"], "fabric_read_bytes_per_propagation": fetch["FETCH_SIZE"] * 1024.0 / props_f,
                  "fabric_read_gbps": fetch["FETCH_SIZE"] * 1024.0 / (ms * 1e-3) / 1e9, "hbm_bytes_per_launch": fetch["FETCH_SIZE"] * 1024.0 + write.get("WRITE_SIZE", 0.0) * 1024.0,
                  "launch_ms_fetch_pass": ms})
        r["hbm_gbps"] = r["hbm_bytes_per_launch"] / (ms * 1e-3) / 1e9
        r["hbm_frac_of_peak"] = r["hbm_gbps"] / 8000.0
    if write.get("WRITE_SIZE") is not None:
        r["WRITE_SIZE_KiB"] = write["WRITE_SIZE"]
    if tcc.get("TCC_HIT_sum") is not None:
        r["tcc_hit_rate"] = tcc["TCC_HIT_sum"] / max(1.0, tcc["TCC_HIT_sum"] + tcc.get("TCC_MISS_sum", 0.0))
        bt = bench_line(f"syn{i}_tcc.log")
        pt = bt.get("balance", {}).get("propagations")
        if pt:
            r["l2_requests_per_propagation"] = (tcc["TCC_HIT_sum"] + tcc.get("TCC_MISS_sum", 0.0)) / pt
    if ea.get("TCC_EA0_RDREQ_sum") is not None:
        be = bench_line(f"syn{i}_ea.log")
        pe = be.get("balance", {}).get("propagations")
        if pe:
            r["ea_read_requests_per_propagation"] = ea["TCC_EA0_RDREQ_sum"] / pe
            r["ea_read_requests_32B_share"] = ea.get("TCC_EA0_RDREQ_32B_sum", 0.0) / max(1.0, ea["TCC_EA0_RDREQ_sum"])
    # issue side of the kernels of stores in global memory (r06): the same SQ passes as for the LDS-resident rows
    sq1, sq2, grbm = (counters(f"syn{i}_{k}") for k in ("sq1", "sq2", "grbm"))
    if sq1 and grbm:
        gui = grbm.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        cap = 1024.0 * gui / 4.0
        bs = bench_line(f"syn{i}_sq1.log")
        ps, ns = bs.get("balance", {}).get("propagations"), bs.get("balance", {}).get("nodes")
        r.update({"valu_busy": sq1["SQ_ACTIVE_INST_VALU"] / cap if cap else None, "salu_busy": sq2.get("SQ_ACTIVE_INST_SCA", 0) / cap if cap else None,
                  "lds_busy": sq2.get("SQ_ACTIVE_INST_LDS", 0) / cap if cap else None,
                  "wait_any_share": sq1["SQ_WAIT_ANY"] / sq1["SQ_WAVE_CYCLES"], "wait_inst_any_share": sq1["SQ_WAIT_INST_ANY"] / sq1["SQ_WAVE_CYCLES"]})
        if ps and ns:
            r.update({"valu_per_64_propagations": sq1["SQ_INSTS_VALU"] / (ps / 64.0), "salu_per_64_propagations": sq1["SQ_INSTS_SALU"] / (ps / 64.0),
                      "valu_per_node": sq1["SQ_INSTS_VALU"] / ns, "salu_per_node": sq1["SQ_INSTS_SALU"] / ns})
    key = "synthetic/" + fp + ("" if i < 3 or i == 5 else ("_beyond_mall" if i == 3 else ("_hot_tier" if team_round else "_no_hot_tier")))
    rec[key] = r
for k, v in earlier.items():
    if k not in rec:
        rec[k] = v
        if isinstance(v, dict):
            v.setdefault("measured", "earlier run of this round (kernel unchanged since)")
json.dump(rec, open(os.path.join(prof, f"{tag}_counters.json"), "w"), indent=1)
print(json.dumps({k: {a: b for a, b in v.items() if a != "counters"} if isinstance(v, dict) else v for k, v in rec.items()}, indent=1))
