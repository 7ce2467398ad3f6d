#!/usr/bin/env python3
"""Where the TIME of a node goes (complements scripts/phase_budget.py, which counts instructions): thread 0's wall clock per workgroup,
summed over the workgroups and divided by the nodes -- the reference's own timers (statistics.hpp:13-29: search, fixpoint, dive) plus,
in the tuning build with knob 0x10000, the inside of the event fixpoint (seeding, rounds; wave 0's core cycles in the rounds split
into record fetch / slice bodies / successor marks / round barrier) and of the branching step (snapshot push, variable selection).
Usage (on the GPU box): TURBO_HIP_LIB=turbo_amd/lib/libturbo_hip_tuning.so python3 scripts/time_budget.py [workload] [nodes] [out.json]
The engine prints the event profile on stderr; this script runs the search in a child process to capture it."""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import os, sys, json
sys.path.insert(0, %r)
from turbo_amd import capi, preprocess
fzn = {"wordpress7_500": "example_wordpress7_500.fzn", "accap_a3": "accap_a3.fzn", "trains15": "trains15.fzn"}[sys.argv[1]]
_, tcn, _ = preprocess.load_fzn_simplified(os.path.join(%r, "benchmarks", fzn))
cfg = capi.make_config(fixpoint=2, stop_after_n_nodes_total=int(sys.argv[2]), timeout_ms=120000, debug=0x10000, verbose=1)
for _ in range(2):
    has, best, st = capi.solve(tcn, cfg)
print("STATS " + json.dumps({k: (list(v) if hasattr(v, "__len__") else v) for k, v in st.items()}))
""" % (ROOT, ROOT)
wl = sys.argv[1] if len(sys.argv) > 1 else "wordpress7_500"
nodes = sys.argv[2] if len(sys.argv) > 2 else "24000000"
out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "r03_time_budget.json")
env = dict(os.environ)
env.setdefault("TURBO_HIP_LIB", os.path.join(ROOT, "turbo_amd", "lib", "libturbo_hip_tuning.so"))
p = subprocess.run([sys.executable, "-c", CHILD, wl, nodes], env=env, capture_output=True, text=True)
m = re.search(r"^STATS (.*)$", p.stdout, re.M)
if not m:
    sys.exit("no statistics: " + (p.stdout + p.stderr)[-1500:])
st = json.loads(m.group(1))
T = st["timers_ns"]
n = st["nodes"]
names = ["overall", "preprocessing", "search", "fixpoint", "transfer_cpu2gpu", "transfer_gpu2cpu", "select_fp_functions", "wait_cpu", "dive"]
per_node = {names[i]: T[i] / n for i in range(len(names))}
# the engine's own phases (tb_stats.prof_ns: r04 had lent them four of the reference's timer keys)
per_node.update({k: v / n for k, v in zip(["fixpoint_seeding", "fixpoint_rounds", "snapshot_push", "variable_selection"], st["prof_ns"])})
prof = [l for l in p.stderr.splitlines() if l.startswith("% event-profile")]
last = {}
for l in prof:  # the second (last) search's lines win
    m = re.search(r"core cycles per node\): fetch (\d+) body (\d+) marks (\d+) barrier (\d+); runs/node ([\d.]+) rounds/node ([\d.]+)", l)
    if m:
        last["wave0_cycles_per_node"] = dict(zip(["record_fetch", "slice_bodies", "successor_marks", "round_barrier"], map(float, m.groups()[:4])))
        last["wave0_runs_per_node"], last["rounds_per_node"] = float(m.group(5)), float(m.group(6))
    m = re.search(r"slices per node ([\d.]+), of wave 0 ([\d.]+)", l)
    if m:
        last["slice_runs_per_node"] = float(m.group(1))
wg_time = per_node["overall"]  # ns of workgroup wall clock per node (= workgroups / nodes-per-second)
fix = per_node["fixpoint"]
acc = {"fixpoint": fix, "search_outside_fixpoint": per_node["search"], "dive": per_node["dive"]}
res = {"workload": wl, "node_budget": int(nodes), "library": os.path.relpath(env["TURBO_HIP_LIB"], ROOT), "nodes": n,
       "nodes_per_sec": n / (st["kernel_ns"] * 1e-9), "workgroups": st["num_blocks"], "threads": st["threads_per_block"],
       "ns_of_workgroup_time_per_node": per_node, "top_level": acc,
       "top_level_attributed": sum(acc.values()) / wg_time if wg_time else None,
       "inside_fixpoint": {"seeding": per_node["fixpoint_seeding"], "rounds": per_node["fixpoint_rounds"],
                           "all_entailed_test_and_exit": fix - per_node["fixpoint_seeding"] - per_node["fixpoint_rounds"]},
       "inside_search": {"snapshot_push": per_node["snapshot_push"], "variable_selection": per_node["variable_selection"],
                         "rest (incumbent, backtrack: restore + replay, bookkeeping by thread 0)": per_node["search"] - per_node["snapshot_push"] - per_node["variable_selection"]},
       "inside_rounds": last,
       "note": "thread 0's wall clock (100 MHz constant clock) summed over workgroups / nodes; wave0_cycles are core-clock cycles of wave 0 inside the rounds. "
               "The tuning build with the timers on runs slower than the production kernel (nodes_per_sec here against the bench line)."}
if "wave0_cycles_per_node" in last:
    c = last["wave0_cycles_per_node"]; tot = sum(c.values())
    res["inside_rounds"]["shares"] = {k: v / tot for k, v in c.items()}
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
