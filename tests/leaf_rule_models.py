"""Small ternary networks for the leaf-rule tests (shared by the CPU and the GPU file): satisfaction problems in which some variable is fixed by no
propagator, so that barebones' rule (a node whose propagators are all entailed is a solution: a BOX, barebones_dive_and_solve.hpp:988-993) and the
gpu / cpu paths' rule (... and the store is extractable: every variable assigned, gpu_dive_and_solve.hpp:333-338, cpu_solving.hpp:33-40) walk different
trees and count different numbers of solutions -- and a brute-force count of the full assignments, independent of oracle/ and of the engine
(tests/rule_brute.py: the constraints themselves)."""
import numpy as np

from rule_brute import holds


def as_tcn(store, props, var_order=1, val_order=0, obj_var=-1):
    """One strategy over the whole store (the reference's default strategy is first_fail / indomain_min over every variable,
    common_solving.hpp:640-650)."""
    from turbo_amd.frontend import TCN
    vo = np.array([var_order], dtype=np.int32)
    vl = np.array([val_order], dtype=np.int32)
    return TCN(store=store, props=props, strat_var_order=vo, strat_val_order=vl, strat_off=np.zeros(2, dtype=np.int32), strat_vars=np.zeros(0, dtype=np.int32),
               obj_var=obj_var)


def loose_network(rng, n_free=None):
    """A few small integer / Boolean variables, a few constraints among SOME of them (reified comparisons whose truth variable nothing else reads,
    sums with a wide result, inequalities that are entailed early) and one to three variables no propagator mentions at all."""
    from turbo_amd.frontend import ITV_DTYPE, PROP_DTYPE
    store = [(0, 0), (1, 1), (2, 2)]
    n = int(rng.integers(3, 6))
    vs = []
    for _ in range(n):
        lo = int(rng.integers(-2, 3))
        store.append((lo, lo + int(rng.integers(1, 4))))
        vs.append(len(store) - 1)
    props = []
    for _ in range(int(rng.integers(1, 5))):
        k = rng.random()
        y, z = int(rng.choice(vs)), int(rng.choice(vs))
        if k < 0.35:    # b = (y <= z), b read by nothing else
            store.append((0, 1)); props.append((7, len(store) - 1, y, z))
        elif k < 0.55:  # b = (y = z)
            store.append((0, 1)); props.append((6, len(store) - 1, y, z))
        elif k < 0.75:  # y <= z
            props.append((7, 1, y, z))
        elif k < 0.9:   # t = y + z, t wide
            store.append((-10, 10)); props.append((0, len(store) - 1, y, z))
        else:           # y != z
            props.append((6, 0, y, z))
    for _ in range(int(rng.integers(1, 4)) if n_free is None else n_free):  # variables nothing constrains
        lo = int(rng.integers(-1, 2))
        store.append((lo, lo + int(rng.integers(1, 3))))
    return np.array(store, dtype=ITV_DTYPE), np.array(props, dtype=PROP_DTYPE)


def brute_force_solutions(store, props):
    """Every full assignment of the box `store` that satisfies every propagator (the definition of the constraints, rule_brute.holds): plain enumeration
    in variable order, a propagator tested as soon as its three variables have a value."""
    n = len(store)
    doms = [range(int(d["lb"]), int(d["ub"]) + 1) for d in store]
    due = [[] for _ in range(n)]
    for p in props:
        op, x, y, z = int(p["op"]), int(p["x"]), int(p["y"]), int(p["z"])
        due[max(x, y, z)].append((op, x, y, z))
    out, a = [], [0] * n

    def rec(v):
        if v == n:
            out.append(tuple(a))
            return
        for val in doms[v]:
            a[v] = val
            if all(holds(op, a[x], a[y], a[z]) for op, x, y, z in due[v]):
                rec(v + 1)
    rec(0)
    assert len(out) <= 200_000
    return out


def box_volume(box):
    w = (box["ub"].astype(np.int64) - box["lb"].astype(np.int64) + 1)
    return int(np.prod(w))
