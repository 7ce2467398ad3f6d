import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import capi, frontend
tcn = frontend.load_fzn(os.path.join(ROOT, "benchmarks", "example_wordpress7_500.fzn"))
G = json.load(open(os.path.join(ROOT, "tests/golden/headline_trees.json")))["example_wordpress7_500.fzn/raw"]["cases"]
for bits in (0x1000000,):
    for case in ("sub0_cut500",):
        rec = G[case]
        has, best, st = capi.solve(tcn, capi.make_config(or_nodes=1, subproblems_power=0, stop_after_n_nodes=rec["cutnodes"], timeout_ms=120000, fixpoint=2, debug=bits, verbose=1))
        print(hex(bits), case, {k: (st[k], rec[k]) for k in ("nodes", "fails", "solutions", "depth_max")}, "why", hex(st["why_not_exhaustive"]), "slice", st["debug_slice"] - 1, flush=True)
