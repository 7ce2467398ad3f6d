import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import frontend, capi
rel = sys.argv[1]
tcn = frontend.load_fzn(os.path.join(ROOT, "benchmarks", rel))
for kw in [dict(fixpoint=2), dict(fixpoint=2, or_nodes=1280), dict(fixpoint=2, or_nodes=1024)]:
    has, best, st = capi.solve(tcn, capi.make_config(timeout_ms=60000, **kw))
    print(kw, {k: st[k] for k in ("nodes","solutions","exhaustive","interrupted","num_blocks","threads_per_block","mem_kind","subproblems_power","eps_solved_subproblems","eps_skipped_subproblems","depth_max","best_bound","why_not_exhaustive")})
