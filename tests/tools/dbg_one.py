#!/usr/bin/env python3
"""Debug aid: solve one instance in several kernel configurations and print what the engine says (why a search was not exhaustive,
self-check findings of the tuning build).  usage: dbg_one.py test_data/pat11.fzn"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from turbo_amd import capi, frontend
rel = sys.argv[1] if len(sys.argv) > 1 else "test_data/pat11.fzn"
tcn = frontend.load_fzn(os.path.join(ROOT, "benchmarks", rel))
print(rel, tcn.n_vars, tcn.n_props)
for name, kw in (("event", dict(fixpoint=2)), ("event_compact", dict(fixpoint=2, debug=0x100000)),
                 ("event_compact_1wg", dict(fixpoint=2, debug=0x100000, or_nodes=1, subproblems_power=0)),
                 ("selfcheck_compact", dict(fixpoint=2, debug=0x100000 | 0x1000000, verbose=1))):
    has, best, st = capi.solve(tcn, capi.make_config(timeout_ms=60000, **kw))
    print(name, has, tcn.objective_of(best) if has else None, {k: st[k] for k in ("exhaustive", "why_not_exhaustive", "debug_slice", "nodes", "fails", "solutions", "num_blocks", "mem_kind", "threads_per_block")}, flush=True)
