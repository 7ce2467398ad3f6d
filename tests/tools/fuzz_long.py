#!/usr/bin/env python3
"""Longer run of the node-level network fuzz of tests/test_gpu_parity.py (seeds beyond the 200 of the test suite), for the compact layouts:
python3 tests/tools/fuzz_long.py [first_seed] [n_seeds]      (GPU box; uses the oracle as the checker)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import pyoracle
from turbo_amd import capi
from fuzz_models import random_network, finite_class_network
first = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n = int(sys.argv[2]) if len(sys.argv) > 2 else 800
COMPACT, C16, C8 = 0x100000, 0x10100000, 0x30100000
MODES = {"event_compact": dict(fixpoint=2, debug=COMPACT), "event_compact16": dict(fixpoint=2, debug=C16), "wac1_compact16": dict(fixpoint=1, debug=C16),
         "event_compact_global": dict(fixpoint=2, debug=COMPACT, only_global_memory=1), "event": dict(fixpoint=2), "event_compact8": dict(fixpoint=2, debug=C8)}
bad = 0
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    # (finite networks: every fourth seed with domains on both sides of COMPACT8's limits -- width 255, bases beyond +-16383)
    store, props = random_network(rng) if seed % 2 == 0 else finite_class_network(rng, scale=(1, 8, 1, 400)[(seed // 2) % 4])[:2]
    stores = [store]
    for _ in range(3):
        s = store.copy()
        for v in rng.choice(np.arange(3, s.shape[0]), size=min(3, s.shape[0] - 3), replace=False):
            lo, hi = int(s["lb"][v]), int(s["ub"][v])
            if lo == capi.TB_NINF or hi == capi.TB_PINF or lo >= hi:
                continue
            m = int(rng.integers(lo, hi + 1))
            if rng.random() < 0.5: s["ub"][v] = m
            else: s["lb"][v] = m
        stores.append(s)
    stores = np.stack(stores)
    exp = [pyoracle.propagate(stores[i], props) for i in range(stores.shape[0])]
    for mode, cfg in MODES.items():
        got, failed, ent, _, _, _ = capi.propagate(props, stores, capi.make_config(timeout_ms=20000, **cfg))
        for i in range(stores.shape[0]):
            e, efailed, eent = exp[i][0], exp[i][1], exp[i][2]
            ok = bool(failed[i]) == efailed and (efailed or (bool(ent[i]) == eent and np.array_equal(got[i], e)))
            if not ok:
                bad += 1
                print(f"MISMATCH seed {seed} mode {mode} store {i}", flush=True)
print(f"seeds {first}..{first + n - 1}: {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
