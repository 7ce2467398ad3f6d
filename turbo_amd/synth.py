"""Synthetic ternary constraint network of BASELINE.json configs[4]:
100 000 integer variables x 500 000 ternary propagators, satisfiable by construction, seed 42.

* 3 interned constants 0, 1, 2 (TCN convention, common_solving.hpp:521);
* `n_base` base variables with domain [0, 9] and a hidden solution s ~ U{0..9};
* `n_def` defined variables `x = y op z`, op in {ADD 70 %, MUL 10 %, MIN 10 %, MAX 10 %}, y and z uniform among
  earlier variables, domain [0, 10^6] (an operator whose hidden value would leave the domain is replaced by MIN);
* side constraints consistent with s: `1 = (y <= z)` (70 %) or `0 = (y = z)` (30 %);
* objective: minimise the last defined variable; strategy input_order / indomain_min over the base variables.

The store is 800 KB, far beyond the 160 KiB of LDS: this is the GLOBAL-memory (HBM/L2 bound) point of the
roofline.  Pure numpy, deterministic.
"""
from __future__ import annotations

import numpy as np

from .frontend import ITV_DTYPE, PROP_DTYPE, TCN

OP_ADD, OP_MUL, OP_MIN, OP_MAX, OP_EQ, OP_LEQ = 0, 1, 4, 5, 6, 7
DOM_MAX = 10 ** 6


def make_synthetic(n_vars: int = 100_000, n_props: int = 500_000, seed: int = 42, base_fraction: float = 0.2) -> TCN:
    rng = np.random.default_rng(seed)
    n_base = int(n_vars * base_fraction)
    n_def = n_vars - n_base
    if n_props < n_def:
        raise ValueError("need at least one propagator per defined variable")
    first = 3  # constants 0, 1, 2
    V = first + n_vars
    store = np.zeros(V, dtype=ITV_DTYPE)
    for c in range(3):
        store[c] = (c, c)
    val = np.zeros(V, dtype=np.int64)
    val[:3] = [0, 1, 2]
    val[first:first + n_base] = rng.integers(0, 10, size=n_base)
    store["lb"][first:first + n_base] = 0
    store["ub"][first:first + n_base] = 9
    props = np.zeros(n_props, dtype=PROP_DTYPE)
    # defined variables (sequential: each one may read the previous ones)
    ops = rng.choice([OP_ADD, OP_MUL, OP_MIN, OP_MAX], size=n_def, p=[0.7, 0.1, 0.1, 0.1])
    u1 = rng.random(n_def)
    u2 = rng.random(n_def)
    for k in range(n_def):
        x = first + n_base + k
        y = first + int(u1[k] * (n_base + k))
        z = first + int(u2[k] * (n_base + k))
        op = int(ops[k])
        a, b = int(val[y]), int(val[z])
        r = a + b if op == OP_ADD else a * b if op == OP_MUL else min(a, b) if op == OP_MIN else max(a, b)
        if r > DOM_MAX:
            op, r = OP_MIN, min(a, b)
        val[x] = r
        props[k] = (op, x, y, z)
    store["lb"][first + n_base:] = 0
    store["ub"][first + n_base:] = DOM_MAX
    # side constraints consistent with the hidden solution
    n_side = n_props - n_def
    ys = first + rng.integers(0, n_vars, size=n_side)
    zs = first + rng.integers(0, n_vars, size=n_side)
    is_leq = rng.random(n_side) < 0.7
    vy, vz = val[ys], val[zs]
    swap = is_leq & (vy > vz)
    ys, zs = np.where(swap, zs, ys), np.where(swap, ys, zs)
    eq_clash = (~is_leq) & (val[ys] == val[zs])  # `y != z` must hold in s: fall back to `y <= z`
    is_leq = is_leq | eq_clash
    side = props[n_def:]
    side["op"] = np.where(is_leq, OP_LEQ, OP_EQ)
    side["x"] = np.where(is_leq, 1, 0)  # constants ONE / ZERO
    side["y"] = ys
    side["z"] = zs
    # interleave side constraints with definitions so that related propagators stay close
    order = np.argsort(np.maximum(np.maximum(props["x"], props["y"]), props["z"]), kind="stable")
    props = props[order]
    base_vars = np.arange(first, first + n_base, dtype=np.int32)
    tcn = TCN(
        store=store, props=np.ascontiguousarray(props),
        strat_var_order=np.array([0, 1], dtype=np.int32),  # input_order, then the default first_fail over the store
        strat_val_order=np.array([0, 0], dtype=np.int32),
        strat_off=np.array([0, n_base, n_base], dtype=np.int32),
        strat_vars=base_vars,
        obj_var=V - 1, goal=1, goal_var=V - 1,
    )
    tcn.hidden_solution = val  # type: ignore[attr-defined]
    return tcn
