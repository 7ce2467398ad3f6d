#!/usr/bin/env python3
"""Debugging aid for tests/test_gpu_parity.py::test_element_models_tree_identical: one seed of fuzz_models.element_model on one workgroup,
node counters against the oracle, under the wake-up filter switches (TB_NO_COND_WAKE / TB_NO_CHAIN_RANGE are read at pack time) and, with the
tuning build (TURBO_HIP_LIB), the self-check of the event fixpoint (knob 0x1000000: which slice could still narrow after a node).
usage: dbg_element_seed.py seed [power] [cut]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1 and sys.argv[1] == "--worker":
    from fuzz_models import element_model
    from turbo_amd import capi, frontend
    seed, power, cut, dbg = (int(a, 0) for a in sys.argv[2:6])
    tcn = frontend.Model.from_string(element_model(seed)).tcn()
    has, best, st = capi.solve(tcn, capi.make_config(or_nodes=1, subproblems_power=power, stop_after_n_nodes=cut, timeout_ms=120000, fixpoint=2, debug=dbg, verbose=1))
    print(json.dumps({k: st[k] for k in ("nodes", "fails", "solutions", "depth_max", "why_not_exhaustive", "debug_slice")}))
    sys.exit(0)
from fuzz_models import element_model
from oracle import pyoracle
from turbo_amd import frontend
seed = int(sys.argv[1]); power = int(sys.argv[2]) if len(sys.argv) > 2 else 0; cut = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
tcn = frontend.Model.from_string(element_model(seed)).tcn()
_, _, st_o = pyoracle.solve(tcn, subproblems_power=power, cutnodes=cut)
print(f"seed {seed} power {power}: {tcn.n_vars} vars {tcn.n_props} props; oracle nodes {st_o['nodes']} fails {st_o['fails']} solutions {st_o['solutions']}")
tuning = os.path.join(ROOT, "turbo_amd", "lib", "libturbo_hip_tuning.so")
for name, env in (("none", dict(TB_NO_COND_WAKE="1", TB_NO_CHAIN_RANGE="1")), ("cond", dict(TB_NO_CHAIN_RANGE="1")), ("range", dict(TB_NO_COND_WAKE="1")), ("both", {})):
    for lib, dbg in ((None, 0x100000), (tuning, 0x100000 | 0x1000000)):
        e = dict(os.environ, **env)
        if lib: e["TURBO_HIP_LIB"] = lib
        p = subprocess.run([sys.executable, __file__, "--worker", str(seed), str(power), str(cut), hex(dbg)], env=e, capture_output=True, text=True, timeout=300)
        tail = [l for l in p.stderr.splitlines() if "self-check" in l][:2]
        print(f"  {name:5s} {'tuning+selfcheck' if lib else 'production     '}: {p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-300:]} {' | '.join(tail)}")
