#!/usr/bin/env python3
"""Instruction budget of the event kernel by phase: SQ_INSTS_VALU / SQ_INSTS_SALU wave-instructions per node.

Every phase of the event kernels is idempotent, so the tuning build can execute ONE selected phase twice (kernels.hpp: reps_of);
the difference in the counters to the undoubled run, divided by the nodes of the run, is that phase's instructions per node.
Runs `rocprofv3 --pmc ... -- python3 scripts/valu_by_phase.py <knob>` once per phase (tuning build, same node budget) and writes
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
VARIANTS = [("base", 0x0), ("base_again", 0x0), ("seeding", 0x100), ("round_scan", 0x200), ("slice_run", 0x300), ("witness", 0x400),
            ("split_scan", 0x500), ("restore_copy", 0x600), ("bitmap_clear", 0x700), ("best_copy", 0x800),
            ("marks", 0x1), ("snapshot_push", 0x4), ("eval_pass", 0x8)]
env = dict(os.environ, TURBO_HIP_LIB=os.path.join(root, "turbo_amd", "lib", "libturbo_hip_tuning.so"), TMPDIR="/tmp")


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
for name, bits in VARIANTS:
    d = os.path.join(out, name)
    subprocess.run(["rm", "-rf", d])
    p = subprocess.run(["rocprofv3", "--pmc", *PMC, "-d", d, "-o", "p", "--", "python3", os.path.join(root, "scripts", "valu_by_phase.py"), hex(bits), workload, nodes],
                       env=env, cwd="/tmp", capture_output=True, text=True)
    m = re.search(r"nodes=(\d+) fails=(\d+) deductions=(\d+) kernel_ns=(\d+)", p.stdout)
    if not m:
        rows[name] = {"error": (p.stdout + p.stderr)[-600:]}
        continue
    n, fails, ded, ns = (int(x) for x in m.groups())
    c = counters(d)
    rows[name] = {"knob": hex(bits), "nodes": n, "fails": fails, "propagations": ded, "kernel_ms": ns / 1e6, "nodes_per_sec": n / (ns * 1e-9),
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
        if name != "eval_pass":  # one evaluation pass is part of a slice run: listed, not added
            sv += dv; ss += ds
    budget["attributed"] = {"valu": sv, "salu": ss, "valu_share": sv / bv, "salu_share": ss / bs}
json.dump({"workload": workload, "node_budget": int(nodes), "counters": PMC, "runs": rows, "per_node_by_phase": budget}, open(os.path.join(root, "gpurun_out", f"{tag}_phase_budget_{workload}.json"), "w"), indent=1)
print(json.dumps(budget, indent=1))
