#!/usr/bin/env python3
"""Instruction budget of a node of the event search kernel by region: static instruction counts of the PRODUCTION-style kernel
between region markers x how often a wave passes each marker.

  static   scripts/region_budget.py static build/h_regions.s      the headline instantiation compiled with -DTB_REGION_MARKERS (kernels.hpp: TB_REGION is an
           assembler comment there; 0.2 % fewer instructions than the production build): every instruction belongs to the last marker before it in layout order
  dynamic  TURBO_HIP_LIB=turbo_amd/lib/libturbo_hip_tuning.so scripts/region_budget.py census [nodes]     (GPU) the tuning build counts marker passes per wave
  table    scripts/region_budget.py table static.json census.json [measured_valu_per_node measured_salu_per_node]

The product is exact for straight-line regions and an approximation where a region holds an inner loop without a marker of its own (block copies, the variable-selection
scan, passes of the rare run kinds) or a per-lane branch; the table states its sum against the measured SQ_INSTS_VALU / SQ_INSTS_SALU per node."""
import collections, json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {0: "kernel prologue / epilogue, subproblem fetch", 1: "node: entry (all waves)", 2: "fixpoint: entry, seeding set-up", 3: "fixpoint: seeding, per 64 changed variables",
         4: "round: scan of the dirty bitmap", 5: "run: prologue (slice info, record fetch, failure check)", 6: "implication run: one pass", 7: "implication run: after the passes",
         8: "implication run: marks", 9: "implication run: conditional wake-up test", 10: "implication run: marks through the adjacency record", 11: "implication run: end",
         12: "other runs: second record, dispatch", 13: "channelling run: prologue", 14: "channelling run: one pass", 15: "channelling run: y written (re-check of the new bounds)",
         16: "channelling run: after the pass", 17: "b = (y ~ k) run", 18: "marks: call", 19: "run: end (counters)", 20: "round: end (flag, barrier)", 21: "fixpoint: after the rounds (bitmap clear)",
         22: "witness test", 23: "witness: rescan of a slice", 24: "fixpoint: exit", 25: "node: thread 0's bookkeeping", 26: "node: after the bookkeeping", 27: "node: best-store copy",
         28: "search loop: top", 29: "search loop: objective bound (thread 0)", 30: "search loop: after the node", 31: "branch: entry", 32: "branch: snapshot push", 33: "variable selection: scan",
         34: "variable selection: thread 0", 35: "branch: decision (thread 0)", 36: "backtrack: restore + replay", 37: "backtrack: thread 0", 38: "subproblem: root restore",
         39: "marks: slots", 40: "marks: adjacency records", 41: "marks: end", 42: "implication run (not lean)", 43: "lean class run", 44: "generic run",
         45: "implication run: marks through the slots", 46: "implication run: marks, end", 47: "channelling run: rest of the pass", 48: "node: exit", 49: "search loop: barrier after the bound",
         50: "branch: before the variable selection", 51: "branch: after the variable selection", 52: "branch: barrier after the decision", 53: "backtrack: barrier", 54: "subproblem: end",
         55: "subproblem: fetch the next one (thread 0)", 56: "subproblem: barrier", 57: "kernel epilogue", 58: "variable selection: barrier", 59: "fixpoint: seeding done",
         60: "dive leaf: subtree skip", 61: "unbounded objective", 62: "node: a solution (thread 0)", 63: "node: node-budget batch (thread 0, every 32 nodes)",
         64: "node: bookkeeping after the leaf test (thread 0)", 65: "node: stop conditions, poll (thread 0)"}


def static(path, pat="solve_kernel"):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if (m := re.match(r"^(_Z\w+):", l)) and pat in m.group(1))
    cur, per = 0, collections.defaultdict(collections.Counter)
    for i in range(start + 1, len(lines)):
        s = lines[i].strip()
        if s.startswith(".Lfunc_end"): break
        m = re.match(r";\s*TBREGION\s+(\d+)", s)
        if m: cur = int(m.group(1)); continue
        if not s or s[0] in ";." or s.endswith(":"): continue
        op = s.split()[0]
        kind = "lane" if op.startswith(("v_readlane", "v_writelane")) else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else \
               "scratch" if op.startswith("scratch_") else "vmem" if op.startswith(("global_", "buffer_", "flat_")) else "other"
        per[cur][kind] += 1
    return {str(k): dict(v) for k, v in sorted(per.items())}


if sys.argv[1] == "static":
    print(json.dumps(static(sys.argv[2]), indent=1))
elif sys.argv[1] == "census":
    sys.path.insert(0, ROOT)
    if len(sys.argv) > 2 and sys.argv[2] == "--worker":
        from turbo_amd import capi, preprocess
        _, tcn, _ = preprocess.load_fzn_simplified(os.path.join(ROOT, "benchmarks", "example_wordpress7_500.fzn"))
        cfg = capi.make_config(fixpoint=2, stop_after_n_nodes_total=int(sys.argv[3]), timeout_ms=120000, verbose=1)
        capi.solve(tcn, cfg)
        sys.exit(0)
    nodes = sys.argv[2] if len(sys.argv) > 2 else "12000000"
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "census", "--worker", nodes], env=dict(os.environ, TB_PRINT_REGIONS="1"), capture_output=True, text=True, timeout=600)
    line = [l for l in p.stderr.splitlines() if l.startswith("% regions")][-1]
    kv = dict(t.split("=") for t in line.split()[2:])
    print(json.dumps({"nodes": int(kv.pop("nodes")), "passes": {k: int(v) for k, v in kv.items() if int(v)}}, indent=1))
else:
    st, ce = json.load(open(sys.argv[2])), json.load(open(sys.argv[3]))
    nodes = ce["nodes"]
    rows, tot = [], collections.Counter()
    for r, c in st.items():
        n = ce["passes"].get(r, 0) if r != "0" else 0
        per_node = n / nodes
        row = {"region": int(r), "what": NAMES.get(int(r), "?"), "wave_passes_per_node": round(per_node, 3), "static": c,
               "valu_per_node": round(per_node * (c.get("valu", 0) + c.get("lane", 0)), 1), "salu_per_node": round(per_node * c.get("salu", 0), 1),
               "lds_per_node": round(per_node * c.get("lds", 0), 2), "vmem_per_node": round(per_node * (c.get("vmem", 0) + c.get("scratch", 0)), 2)}
        rows.append(row)
        for k in ("valu_per_node", "salu_per_node", "lds_per_node", "vmem_per_node"): tot[k] += row[k]
    rows.sort(key=lambda r: -r["valu_per_node"])
    out = {"what": __doc__.strip(), "nodes_of_the_census": nodes, "sum": {k: round(v, 1) for k, v in tot.items()}, "regions": rows}
    if len(sys.argv) > 5:
        mv, ms = float(sys.argv[4]), float(sys.argv[5])
        out["measured"] = {"SQ_INSTS_VALU_per_node": mv, "SQ_INSTS_SALU_per_node": ms, "valu_attributed": round(tot["valu_per_node"] / mv, 3), "salu_attributed": round(tot["salu_per_node"] / ms, 3),
                           "note": "SQ_INSTS_VALU counts v_readlane / v_writelane as VALU: the table does too"}
    print(json.dumps(out, indent=1))
