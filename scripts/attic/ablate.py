# needs the tuning build of the engine: make hip EXTRA_HIPFLAGS=-DTB_TUNING (the knobs are compiled out otherwise)
"""Ablation of the sweep on tb_propagate (profiling knobs in tb_config.reserved[0]); results are NOT valid fixpoints."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from turbo_amd import frontend, capi
name = sys.argv[1] if len(sys.argv) > 1 else "example_wordpress7_500.fzn"
n_stores = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
sweeps = 40
tcn = frontend.load_fzn(os.path.join(ROOT, "benchmarks", name))
stores = np.repeat(tcn.store[None, :], n_stores, axis=0)
for fp in (0, 1):
    for dbg, label in [(0, "full"), (1, "no evaluate"), (2, "no gathers"), (3, "no eval+gathers"), (4, "no tail"), (8, "no bytecode loads"), (15, "nothing but loop")]:
        cfg = capi.make_config(fixpoint=fp)
        cfg.reserved[0] = dbg | (sweeps << 8)
        best = 1e30
        for _ in range(2):
            out, failed, ent, iters, ded, ns = capi.propagate(tcn.props, stores, cfg)
            best = min(best, ns)
        props = n_stores * sweeps * tcn.n_props
        print(f"{name} fp={'wac1' if fp else 'ac1'} {label:20s}: {best*1e-6:8.2f} ms  {props/(best*1e-9):.3e} props/s (forced {sweeps} sweeps)", flush=True)
