"""Micro-benchmark of the inner loop (tb_propagate = block fixpoint + entailment on a batch of search nodes).
usage: sweep_bench.py [instance.fzn] [n_stores] [reps] [key=value tb_config overrides...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from turbo_amd import frontend, capi

name = sys.argv[1] if len(sys.argv) > 1 else "example_wordpress7_500.fzn"
n_stores = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
over = dict(kv.split("=") for kv in sys.argv[4:])
over = {k: int(v) for k, v in over.items()}
tcn = frontend.load_fzn(os.path.join(ROOT, "benchmarks", name))
rng = np.random.default_rng(1)
root, failed, ent, it, de, ns = capi.propagate(tcn.props, tcn.store[None, :], capi.make_config(**over))
root = root[0]
stores = np.repeat(root[None, :], n_stores, axis=0)
# each store: root fixpoint + a few random (unpropagated) decisions on the first search strategy's variables
svars = tcn.strat_vars[tcn.strat_off[0]:tcn.strat_off[1]] if tcn.n_strats > 1 else np.arange(tcn.n_vars)
for s in range(n_stores):
    for _ in range(int(rng.integers(1, 6))):
        v = int(rng.choice(svars))
        lo, hi = int(stores[s]["lb"][v]), int(stores[s]["ub"][v])
        if lo < hi:
            mid = int(rng.integers(lo, hi + 1))
            if rng.random() < 0.5: stores[s]["ub"][v] = mid
            else: stores[s]["lb"][v] = mid
for label, kw in [("wac1", dict(fixpoint=1)), ("ac1", dict(fixpoint=0))]:
    cfg = capi.make_config(**{**kw, **over})
    best = None
    for _ in range(reps):
        out, failed, ent, iters, ded, ns = capi.propagate(tcn.props, stores, cfg)
        rate = ded.sum() / (ns * 1e-9)
        best = max(best or 0, rate)
    print(f"{name} {label} {over}: stores={n_stores} failed={int((failed!=0).sum())} sweeps/store={iters.mean():.1f} "
          f"props={int(ded.sum())} kernel={ns*1e-6:.2f} ms  -> {best:.3e} props/s  ({best*40/1e12:.2f} TB/s algorithmic)", flush=True)
