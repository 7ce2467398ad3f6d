#!/usr/bin/env python3
"""Root-domain widths of the integer variables of a simplified instance: python3 scripts/r04_width_hist.py [instance ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import preprocess
for name in sys.argv[1:] or ["trains15.fzn"]:
    _, tcn, _ = preprocess.load_fzn_simplified(os.path.join(ROOT, "benchmarks", name))
    st = np.asarray(tcn.store)
    lb, ub = st['lb'].astype(np.int64), st['ub'].astype(np.int64)
    const = lb == ub
    boolean = (lb >= 0) & (ub <= 1) & ~const
    ints = ~const & ~boolean
    w = (ub - lb)[ints]
    print(f"{name}: {len(lb)} variables, {const.sum()} constants, {boolean.sum()} Booleans, {ints.sum()} integers")
    for lim in (15, 63, 127, 255, 1023, 4095, 65535):
        print(f"   width <= {lim}: {(w <= lim).sum()}")
    print("   absolute values within -128..127:", ((lb[ints] >= -128) & (ub[ints] <= 127)).sum(), " within 0..255:", ((lb[ints] >= 0) & (ub[ints] <= 255)).sum())
    print("   lb min/max", lb[ints].min(), lb[ints].max(), "ub min/max", ub[ints].min(), ub[ints].max())
