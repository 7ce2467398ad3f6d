#!/usr/bin/env python3
"""Instruction budget of the event kernel by phase: SQ_INSTS_VALU / SQ_INSTS_SALU wave-instructions per node.

Every phase of the event kernels is idempotent, so the tuning build can execute ONE selected phase twice (kernels.hpp: reps_of);
the difference in the counters to the undoubled run, divided by the nodes of the run, is that phase's instructions per node.
Runs `rocprofv3 --pmc ... -- python3 scripts/valu_by_phase.py` once per phase, each time with the library that executes that phase twice
(`make phases`: production kernels, the phase chosen at compile time; same node budget) and writes
gpurun_out/<tag>_phase_budget.json.  Usage (on the GPU box): python3 scripts/phase_budget.py [tag] [workload] [nodes]
This process never touches the GPU; the profiled children do."""
import glob, json, os, re, sqlite3, subprocess, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
workload = sys.argv[2] if len(sys.argv) > 2 else "wordpress7_500"
nodes = sys.argv[3] if len(sys.argv) > 3 else "12000000"
out = os.path.join(root, "gpurun_out", f"{tag}_phase_{workload}")
os.makedirs(out, exist_ok=True)
PMC = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU"]
# (name, library): the production library twice (noise), then one library per doubled phase (make phases)
VARIANTS = [("base", None), ("base_again", None), ("seeding", 1), ("round_scan", 2), ("slice_run", 3), ("witness", 4),
            ("split_scan", 5), ("restore_copy", 6), ("bitmap_clear", 7), ("best_copy", 8), ("run_prologue", 9), ("round_end", 10),
            ("marks", 11), ("snapshot_push", 12), ("eval_pass", 13), ("count_runs", 14)]
libdir = os.path.join(root, "turbo_amd", "lib")


def counters(d):
    dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
    if not dbs:
        return {}
    con = sqlite3.connect(dbs[0])
    res = {}
    for name, total in con.execute("select counter_name, sum(value) from counters_collection where kernel_name like '%solve_kernel%' group by counter_name"):
        res[name] = total
    return res


rows = {}
for name, phase in VARIANTS:
    d = os.path.join(out, name)
    subprocess.run(["rm", "-rf", d])
    lib = os.path.join(libdir, "libturbo_hip.so") if phase is None else os.path.join(libdir, "phases", f"phase_{phase}.so")
    if not os.path.exists(lib):
        rows[name] = {"error": f"{lib} is not built (make phases)"}
        continue
    bits = 0
    env = dict(os.environ, TURBO_HIP_LIB=lib, TMPDIR="/tmp")
    p = subprocess.run(["rocprofv3", "--pmc", *PMC, "-d", d, "-o", "p", "--", "python3", os.path.join(root, "scripts", "valu_by_phase.py"), hex(bits), workload, nodes],
                       env=env, cwd="/tmp", capture_output=True, text=True)
    m = re.search(r"nodes=(\d+) fails=(\d+) deductions=(\d+) kernel_ns=(\d+)", p.stdout)
    if not m:
        rows[name] = {"error": (p.stdout + p.stderr)[-600:]}
        continue
    n, fails, ded, ns = (int(x) for x in m.groups())
    c = counters(d)
    rows[name] = {"library": os.path.relpath(lib, root), "nodes": n, "fails": fails, "propagations": ded, "kernel_ms": ns / 1e6, "nodes_per_sec": n / (ns * 1e-9),
                  **{k.lower() + "_per_node": v / n for k, v in c.items()}}
    print(name, rows[name], flush=True)
base = rows.get("base", {})
budget = {}
if "sq_insts_valu_per_node" in base:
    b2 = rows.get("base_again", base)
    bv = (base["sq_insts_valu_per_node"] + b2.get("sq_insts_valu_per_node", base["sq_insts_valu_per_node"])) / 2
    bs = (base["sq_insts_salu_per_node"] + b2.get("sq_insts_salu_per_node", base["sq_insts_salu_per_node"])) / 2
    budget["total"] = {"valu": bv, "salu": bs, "noise_valu": abs(base["sq_insts_valu_per_node"] - b2.get("sq_insts_valu_per_node", 0)),
                       "noise_salu": abs(base["sq_insts_salu_per_node"] - b2.get("sq_insts_salu_per_node", 0))}
    sv = ss = 0.0
    for name, _ in VARIANTS[2:]:
        r = rows.get(name, {})
        if "sq_insts_valu_per_node" not in r:
            continue
        dv, ds = r["sq_insts_valu_per_node"] - bv, r["sq_insts_salu_per_node"] - bs
        budget[name] = {"valu": dv, "salu": ds, "valu_share": dv / bv, "salu_share": ds / bs}
        if name == "count_runs":  # same kernel, `propagations` then counts slice runs x 64 instead of wave iterations x 64
            budget.pop(name)
            continue
        if name != "eval_pass":  # one evaluation pass is part of a slice run: listed, not added
            sv += dv; ss += ds
    # passes beyond the first of a run (a run iterates its slice to a local fixpoint): (wave iterations - runs) x one evaluation pass
    if "count_runs" in rows and "propagations" in rows["count_runs"] and "eval_pass" in budget:
        runs = rows["count_runs"]["propagations"] / 64.0 / rows["count_runs"]["nodes"]
        iters = base["propagations"] / 64.0 / base["nodes"]
        extra = max(0.0, iters - runs)
        per_run_v, per_run_s = budget["eval_pass"]["valu"] / max(runs, 1e-9), budget["eval_pass"]["salu"] / max(runs, 1e-9)
        budget["further_passes"] = {"valu": extra * per_run_v, "salu": extra * per_run_s, "valu_share": extra * per_run_v / bv, "salu_share": extra * per_run_s / bs,
                                    "note": f"{runs:.1f} slice runs and {iters:.1f} wave iterations per node: {extra:.1f} further passes x one evaluation pass (the write tail of a narrowing pass is not in this figure)"}
        sv += budget["further_passes"]["valu"]; ss += budget["further_passes"]["salu"]
    budget["attributed"] = {"valu": sv, "salu": ss, "valu_share": sv / bv, "salu_share": ss / bs,
                            "not_doubled": "slice iteration (next set bit of the dirty words), the write tail of narrowing passes, thread 0's node bookkeeping (statistics, clocks, "
                                           "stop conditions, incumbent), decisions / objective tightening / backtracking by thread 0, loop control of the search loop, subproblem fetch"}
json.dump({"workload": workload, "node_budget": int(nodes), "counters": PMC, "runs": rows, "per_node_by_phase": budget}, open(os.path.join(root, "gpurun_out", f"{tag}_phase_budget_{workload}.json"), "w"), indent=1)
print(json.dumps(budget, indent=1))
