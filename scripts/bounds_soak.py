#!/usr/bin/env python3
"""Soak of the software bounds build (`make bounds`: kernels.hpp TB_BOUNDS, every index the kernels form from host-packed fields is range checked and a
violation comes back as TB_ERR_HIP "bounds build: ... site S, index I, limit L, workgroup W" instead of a memory fault).

    TURBO_HIP_LIB=turbo_amd/lib/libturbo_hip_bounds.so python3 scripts/bounds_soak.py [seconds per configuration] [out.json]

Runs, on the bounds library, the search configurations bench.py launches (the three instances in the engine's own plans, the synthetic 100k x 500k network in
both fixpoints, plus the layouts the planner does not pick by itself: COMPACT16, slabs in global memory, four-wave workgroups, the sweeps) for a wall-clock
budget each, in full-grid launches with a node budget, and sums the nodes per store layout.  Any report ends the run with the site.  The fuzz families and
the two-rank suite run on the same library through pytest (scripts/bounds_soak.sh)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TURBO_HIP_LIB", os.path.join(ROOT, "turbo_amd", "lib", "libturbo_hip_bounds.so"))
from turbo_amd import capi, preprocess  # noqa: E402
from turbo_amd.synth import make_synthetic  # noqa: E402

budget_s = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "r06_bounds_soak.json")
COMPACT, C16, C8 = 0x100000, 0x10100000, 0x30100000
LAYOUT = {0: "plain", 1: "COMPACT", 2: "COMPACT16", 3: "HOT", 4: "COMPACT8", 5: "TEAM"}

nets = {}
for name in ("example_wordpress7_500.fzn", "accap_a3.fzn", "trains15.fzn"):
    nets[name] = preprocess.load_fzn_simplified(os.path.join(ROOT, "benchmarks", name))[1]
nets["synthetic"] = make_synthetic(100000, 500000, seed=42)

CASES = [  # (network, label, config, nodes per launch)
    ("example_wordpress7_500.fzn", "event (engine's plan)", dict(fixpoint=2), 40_000_000),
    ("example_wordpress7_500.fzn", "event, gpu leaf rule", dict(fixpoint=2, leaf_requires_assignment=1), 40_000_000),
    ("example_wordpress7_500.fzn", "event, four-wave workgroups", dict(fixpoint=2, threads_per_block=256), 20_000_000),
    ("example_wordpress7_500.fzn", "event, compact slab in global memory", dict(fixpoint=2, only_global_memory=1, debug=COMPACT), 10_000_000),
    ("example_wordpress7_500.fzn", "event, plain store", dict(fixpoint=2, debug=0x80000), 3_000_000),
    ("example_wordpress7_500.fzn", "wac1", dict(fixpoint=1), 2_000_000),
    ("example_wordpress7_500.fzn", "ac1", dict(fixpoint=0), 400_000),
    ("example_wordpress7_500.fzn", "wac1 + entailed removal", dict(fixpoint=1, entailed_prop_removal=1), 2_000_000),
    ("accap_a3.fzn", "event (engine's plan)", dict(fixpoint=2), 60_000_000),
    ("accap_a3.fzn", "event, COMPACT16 forced", dict(fixpoint=2, debug=C16), 40_000_000),
    ("accap_a3.fzn", "event, COMPACT8 forced", dict(fixpoint=2, debug=C8), 40_000_000),
    ("accap_a3.fzn", "wac1", dict(fixpoint=1), 40_000_000),
    ("trains15.fzn", "event (engine's plan: COMPACT8)", dict(fixpoint=2), 30_000_000),
    ("trains15.fzn", "event, COMPACT16", dict(fixpoint=2, debug=C16 | 0x80000000), 20_000_000),
    ("trains15.fzn", "event, COMPACT", dict(fixpoint=2, debug=COMPACT | 0x20000000), 10_000_000),
    ("trains15.fzn", "event, COMPACT8 slab in global memory", dict(fixpoint=2, only_global_memory=1, debug=C8), 10_000_000),
    ("trains15.fzn", "wac1", dict(fixpoint=1), 5_000_000),
    ("synthetic", "event (hot tier)", dict(fixpoint=2), 30_000),
    ("synthetic", "wac1 (workgroup teams, the default plan)", dict(fixpoint=1), 15_000),
]

rows, failed = [], None
for net, label, cfg, per_launch in CASES:
    tcn = nets[net]
    t0 = time.time()
    nodes = launches = 0
    plan = None
    while time.time() - t0 < budget_s:
        try:
            s = capi.Session(tcn, capi.make_config(stop_after_n_nodes_total=per_launch, timeout_ms=600000, **cfg))
            plan = s.plan()
            s.start()
            while not s.poll()[1]:
                time.sleep(0.002)
            has, best, st = s.finish()
            s.close()
        except capi.TurboHipError as e:
            failed = f"{net} / {label}: {e}"
            break
        nodes += st["nodes"]
        launches += 1
    row = {"network": net, "configuration": label, "launches": launches, "nodes": nodes, "seconds": round(time.time() - t0, 1),
           "layout": LAYOUT.get(plan["kernel_opt"] if plan and plan["kernel_event"] else (plan["kernel_opt"] >> 1 if plan else 0), "?") if plan else None,
           "plan": plan}
    rows.append(row)
    print(json.dumps({k: v for k, v in row.items() if k != "plan"}), flush=True)
    if failed:
        break
by_layout = {}
for r in rows:
    by_layout[r["layout"]] = by_layout.get(r["layout"], 0) + r["nodes"]
res = {"library": os.path.relpath(os.environ["TURBO_HIP_LIB"], ROOT), "seconds_per_configuration": budget_s, "bounds_report": failed, "hits": 0 if failed is None else 1,
       "nodes_by_layout": by_layout, "rows": rows}
os.makedirs(os.path.dirname(out_path), exist_ok=True)
with open(out_path, "w") as f:
    json.dump(res, f, indent=1)
print("BOUNDS SOAK:", "no report" if failed is None else failed, json.dumps(by_layout))
sys.exit(0 if failed is None else 1)
