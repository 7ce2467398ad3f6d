#!/usr/bin/env python3
"""Same-box census of the event fixpoint under the two r04 wake-up filters (engine.hip: Chains / conditional wake-up; switched off at pack time by
TB_NO_CHAIN_RANGE / TB_NO_COND_WAKE): slice runs per node by class and how many of them narrowed nothing (tuning build, knobs 0x400000 / 0x40 /
bits 28-31), wave 0's marks (runs that marked something, lanes that did), and nodes/s of the production build -- the evidence behind "what a node
wastes is whole runs that narrow nothing, not idle lanes of runs that do" (DESIGN.md section 7).
usage (GPU box): python3 scripts/r04_census_ab.py [instance] > profiles/r04_wakeup_filters_ab.json"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CLASSES = ["heavy", "add", "min", "max", "eq_r", "leq_r", "eq_t", "eq_f", "leq_t", "leq_f", "mixed"]
if len(sys.argv) > 1 and sys.argv[1] == "--worker":
    from turbo_amd import capi, preprocess
    name, tuning = sys.argv[2], sys.argv[3] == "1"
    _, tcn, _ = preprocess.load_fzn_simplified(os.path.join(ROOT, "benchmarks", name))

    def run(bits, nodes=3_000_000, verbose=0):
        cfg = capi.make_config(fixpoint=2, stop_after_n_nodes_total=nodes, timeout_ms=120000, debug=bits, verbose=verbose)
        capi.solve(tcn, cfg)
        _, _, st = capi.solve(tcn, cfg)
        return st
    out = {}
    if tuning:
        st = run(0x400000)
        out["runs_per_node"] = st["num_deductions"] / 64.0 / st["nodes"]
        out["useless_runs_per_node"] = run(0x400040)["num_deductions"] / 64.0 / st["nodes"]
        st = run(0)
        out["wave_iterations_per_node"] = st["num_deductions"] / 64.0 / st["nodes"]
        out["active_lane_evaluations_per_node"] = st["active_lane_evaluations"] / st["nodes"]
        out["by_class"] = {}
        for c, cname in enumerate(CLASSES):
            a = run(0x400000 | ((c + 1) << 28))
            r = a["num_deductions"] / 64.0 / a["nodes"]
            if r > 0.05:
                u = run(0x400040 | ((c + 1) << 28))
                out["by_class"][cname] = {"runs_per_node": r, "narrowed_nothing": u["num_deductions"] / 64.0 / u["nodes"]}
        run(0x10000, 6_000_000, verbose=1)  # wave 0's time / marks census goes to stderr
    else:
        st = run(0, 48_000_000)
        out["nodes_per_sec"] = st["nodes"] / (st["kernel_ns"] * 1e-9)
        out["evaluations_per_node_x64"] = st["num_deductions"] / st["nodes"]
        out["active_lane_evaluations_per_node"] = st["active_lane_evaluations"] / st["nodes"]
        out["workgroups"] = st["num_blocks"]
    print("RESULT " + json.dumps(out))
    sys.exit(0)
name = sys.argv[1] if len(sys.argv) > 1 else "example_wordpress7_500.fzn"
tuning_lib = os.path.join(ROOT, "turbo_amd", "lib", "libturbo_hip_tuning.so")
rec = {"what": __doc__.split("usage")[0].strip(), "instance": name, "variants": {}}
for v, env in (("no filter (r03 behaviour)", dict(TB_NO_COND_WAKE="1", TB_NO_CHAIN_RANGE="1")), ("conditional wake-up only", dict(TB_NO_CHAIN_RANGE="1")),
               ("chain slices by value range only", dict(TB_NO_COND_WAKE="1")), ("both (production)", {})):
    row = {}
    for tuning in (True, False):
        e = dict(os.environ, **env)
        if tuning:
            e["TURBO_HIP_LIB"] = tuning_lib
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", name, "1" if tuning else "0"], env=e, capture_output=True, text=True, timeout=900)
        res = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
        if not res:
            row["error"] = p.stderr[-500:]
            continue
        row["tuning_build" if tuning else "production_build"] = json.loads(res[-1][7:])
        if tuning:
            prof = [l[2:].strip() for l in p.stderr.splitlines() if l.startswith("% event-profile")]
            row["tuning_build"]["wave0_profile"] = prof[-6:]
            m = re.search(r"runs with marks ([\d.]+) \(([\d.]+) lanes\)", "\n".join(prof[-6:]))
            if m:
                row["tuning_build"]["wave0_runs_with_marks_per_node"] = float(m.group(1))
                row["tuning_build"]["wave0_lanes_marking_per_node"] = float(m.group(2))
                row["tuning_build"]["lanes_per_marking_run"] = float(m.group(2)) / max(1e-9, float(m.group(1)))
    rec["variants"][v] = row
print(json.dumps(rec, indent=1))
