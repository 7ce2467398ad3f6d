#!/usr/bin/env python3
"""How much work does a search node really need?  CPU-side census used to design the event-driven kernels.

Walks a DFS path of an instance (left branches, then right branches on the way back) and, per node, counts
  * the propagators that must be re-evaluated with a PROPAGATOR-level worklist (one FIFO entry per propagator reading a
    narrowed variable),
  * the 64-propagator slices a SLICE-level worklist runs (the engine's unit: class-sorted records, a slice is evaluated
    whole and iterated to its local fixpoint),
so that the cost of slice granularity (evaluations per useful evaluation) and the spread of a variable's readers over
slices are visible.  Test infrastructure: it drives oracle/ (orc_deduce) and is not part of the product.

usage: python tests/tools/event_profile.py [--nodes 120] [--order engine|clustered] instance.fzn
"""
import argparse
import collections
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402
from turbo_amd import frontend, preprocess  # noqa: E402

K_HEAVY, K_ADD, K_MIN, K_MAX, K_EQ_R, K_LEQ_R, K_EQ_T, K_EQ_F, K_LEQ_T, K_LEQ_F = range(10)


def class_of(op, xc, xv):
    if op == 0: return K_ADD
    if op == 4: return K_MIN
    if op == 5: return K_MAX
    if op == 6: return (K_EQ_T if xv >= 1 else K_EQ_F) if xc else K_EQ_R
    if op == 7: return (K_LEQ_T if xv >= 1 else K_LEQ_F) if xc else K_LEQ_R
    return K_HEAVY


def engine_order(tcn):
    st = tcn.store
    keys = []
    for p in tcn.props:
        d = st[p["x"]]
        xc = d["lb"] == d["ub"] and d["lb"] not in (-2**31, 2**31 - 1)
        keys.append(class_of(int(p["op"]), xc, int(d["lb"])) * 16 + int(p["op"]))
    return np.argsort(np.array(keys), kind="stable")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=120)
    ap.add_argument("--raw", action="store_true")
    ap.add_argument("instance")
    a = ap.parse_args()

    def oprop(store, props):
        out, failed, _, _, _ = pyoracle.propagate(store, props)
        return out, failed
    path = a.instance if os.path.exists(a.instance) else os.path.join(ROOT, "benchmarks", a.instance)
    tcn = frontend.load_fzn(path) if a.raw else preprocess.load_fzn_simplified(path, propagate=oprop)[1]
    order = engine_order(tcn)
    props = np.ascontiguousarray(tcn.props[order])
    P, V = props.shape[0], tcn.n_vars
    L = pyoracle.lib()
    is_const = (tcn.store["lb"] == tcn.store["ub"])
    readers = [[] for _ in range(V)]
    for i, p in enumerate(props):
        for v in {int(p["x"]), int(p["y"]), int(p["z"])}:
            if not is_const[v]:
                readers[v].append(i)
    slices_of = [sorted({i // 64 for i in r}) for r in readers]
    deg = np.array([len(r) for r in readers])
    sdeg = np.array([len(s) for s in slices_of])
    live = ~is_const
    print(f"{os.path.basename(path)}: V={V} P={P} slices={(P + 63) // 64}; readers per variable: mean {deg[live].mean():.1f} max {deg.max()}; "
          f"slices per variable: mean {sdeg[live].mean():.2f} max {sdeg.max()}; ideal slices per variable (readers/64 rounded up): {np.ceil(deg[live] / 64).mean():.2f}")

    base = props.ctypes.data
    f = C.c_int(0)

    def run_props(store, changed_vars):
        """propagator-level worklist; returns (evaluations, failed)"""
        q = collections.deque()
        inq = np.zeros(P, dtype=bool)
        for v in changed_vars:
            for i in readers[v]:
                if not inq[i]:
                    inq[i] = True; q.append(i)
        evals = 0
        sp = store.ctypes.data
        while q:
            i = q.popleft(); inq[i] = False
            p = props[i]
            x, y, z = int(p["x"]), int(p["y"]), int(p["z"])
            before = (store[x].copy(), store[y].copy(), store[z].copy())
            f.value = 0
            L.orc_deduce(base + 16 * i, sp, C.byref(f))
            evals += 1
            if f.value:
                return evals, True
            for v, b in zip((x, y, z), before):
                if store[v] != b:
                    for j in readers[v]:
                        if j != i and not inq[j]:
                            inq[j] = True; q.append(j)
                    if not inq[i]:  # a propagator is not idempotent in general: look at it again
                        inq[i] = True; q.append(i)
        return evals, False

    def run_slices(store, changed_vars):
        """slice-level worklist with wave-local iteration; returns (slice runs, evaluations = 64 x iterations, failed)"""
        S = (P + 63) // 64
        dirty = np.zeros(S, dtype=bool)
        q = collections.deque()
        for v in changed_vars:
            for s in slices_of[v]:
                if not dirty[s]:
                    dirty[s] = True; q.append(s)
        runs = evals = 0
        sp = store.ctypes.data
        while q:
            s = q.popleft(); dirty[s] = False
            runs += 1
            lo, hi = s * 64, min(P, s * 64 + 64)
            while True:
                evals += 64
                changed_here = set()
                for i in range(lo, hi):
                    p = props[i]
                    x, y, z = int(p["x"]), int(p["y"]), int(p["z"])
                    before = (store[x].copy(), store[y].copy(), store[z].copy())
                    f.value = 0
                    L.orc_deduce(base + 16 * i, sp, C.byref(f))
                    if f.value:
                        return runs, evals, True
                    for v, b in zip((x, y, z), before):
                        if store[v] != b:
                            changed_here.add(v)
                if not changed_here:
                    break
                for v in changed_here:
                    for t in slices_of[v]:
                        if t != s and not dirty[t]:
                            dirty[t] = True; q.append(t)
        return runs, evals, False

    # which (operand, bound) events can enable new narrowing by a propagator: everything, except for the comparisons whose
    # truth variable is a constant -- y <= z only reacts to y.lb / z.ub, y > z only to y.ub / z.lb
    root, failed, _, _, _ = pyoracle.propagate(tcn.store, tcn.props)
    assert not failed
    xconst = np.array([tcn.store["lb"][int(p["x"])] == tcn.store["ub"][int(p["x"])] for p in props])
    rel_lb = [set() for _ in range(V)]  # slices interested in a raise of v.lb
    rel_ub = [set() for _ in range(V)]
    for i, p in enumerate(props):
        s_ = i // 64
        op, x, y, z = int(p["op"]), int(p["x"]), int(p["y"]), int(p["z"])
        if op == 7 and xconst[i]:
            if tcn.store["lb"][x] >= 1:
                rel_lb[y].add(s_); rel_ub[z].add(s_)
            else:
                rel_ub[y].add(s_); rel_lb[z].add(s_)
        else:
            for v in (x, y, z):
                rel_lb[v].add(s_); rel_ub[v].add(s_)
    S = (P + 63) // 64
    names = ["HEAVY", "ADD", "MIN", "MAX", "EQ_R", "LEQ_R", "EQ_T", "EQ_F", "LEQ_T", "LEQ_F", "mixed"]
    slice_class = []
    for s_ in range(S):
        cl = set()
        for i in range(s_ * 64, min(P, s_ * 64 + 64)):
            d = tcn.store[int(props[i]["x"])]
            cl.add(class_of(int(props[i]["op"]), bool(d["lb"] == d["ub"]), int(d["lb"])))
        slice_class.append(cl.pop() if len(cl) == 1 else 10)
    iter_hist = collections.defaultdict(lambda: [0, 0])  # class -> [iterations, runs]

    def run_slices2(store, seeds, filtered):
        """seeds: list of (var, lb_changed, ub_changed).  Returns (runs, useful runs, evaluations, failed, slices with only irrelevant events)."""
        dirty = np.zeros(S, dtype=bool)
        entd = np.zeros(S, dtype=bool)
        q = collections.deque()

        def mark(v, lbc, ubc, me):
            for t in slices_of[v]:
                if t == me:
                    continue
                relevant = (not filtered) or (lbc and t in rel_lb[v]) or (ubc and t in rel_ub[v])
                if relevant:
                    if not dirty[t]:
                        dirty[t] = True; q.append(t)
                else:
                    entd[t] = True
        for v, lbc, ubc in seeds:
            mark(v, lbc, ubc, -1)
        runs = useful = evals = 0
        sp = store.ctypes.data
        while q:
            s_ = q.popleft(); dirty[s_] = False; entd[s_] = False
            runs += 1
            if not filtered:
                iter_hist[slice_class[s_]][1] += 1
            lo, hi = s_ * 64, min(P, s_ * 64 + 64)
            first = True
            while True:
                evals += 64
                if not filtered:
                    iter_hist[slice_class[s_]][0] += 1
                ch = {}
                for i in range(lo, hi):
                    p = props[i]
                    x, y, z = int(p["x"]), int(p["y"]), int(p["z"])
                    b = [(int(store["lb"][v]), int(store["ub"][v])) for v in (x, y, z)]
                    f.value = 0
                    L.orc_deduce(base + 16 * i, sp, C.byref(f))
                    if f.value:
                        return runs, useful, evals, True, int(entd.sum())
                    for v, (l0, u0) in zip((x, y, z), b):
                        l1, u1 = int(store["lb"][v]), int(store["ub"][v])
                        if l1 != l0 or u1 != u0:
                            a0, a1 = ch.get(v, (False, False))
                            ch[v] = (a0 or l1 != l0, a1 or u1 != u0)
                if not ch:
                    break
                if first:
                    useful += 1; first = False
                for v, (lbc, ubc) in ch.items():
                    mark(v, lbc, ubc, s_)
        return runs, useful, evals, False, int(entd.sum())

    svars = tcn.strat_vars[tcn.strat_off[0]:tcn.strat_off[1]] if tcn.strat_off[1] > tcn.strat_off[0] else np.arange(V)
    vo = int(tcn.strat_var_order[0])
    obj = int(tcn.obj_var)

    def pick(store):
        best, bv = None, -1
        for v in svars:
            lb, ub = int(store["lb"][v]), int(store["ub"][v])
            if lb == ub or lb == -2**31 or ub == 2**31 - 1:
                continue
            key = (ub - lb) if vo == 1 else (lb if vo == 3 else 0)
            if best is None or key < best:
                best, bv = key, int(v)
                if vo == 0:
                    break
        return bv

    rows = []
    best_obj = None
    stack = []
    cur = root.copy()

    def node(store, v, is_left, lb):
        seeds = []
        if is_left:
            store["ub"][v] = lb; seeds.append((v, False, True))
        else:
            store["lb"][v] = lb + 1; seeds.append((v, True, False))
        if best_obj is not None and obj >= 0 and int(store["ub"][obj]) > best_obj - 1:
            store["ub"][obj] = best_obj - 1; seeds.append((obj, False, True))
        s2 = store.copy()
        r0 = run_slices2(store, seeds, False)
        r1 = run_slices2(s2, seeds, True)
        assert r0[3] == r1[3] and (r0[3] or (store == s2).all()), "filtered marks changed the fixpoint"
        rows.append((r0[0], r0[1], r0[2], r1[0], r1[1], r1[2], r1[4], r0[3], best_obj is not None))
        return r0[3]

    while len(rows) < a.nodes:
        v = pick(cur)
        if v < 0:  # a solution leaf
            if obj >= 0:
                best_obj = int(cur["lb"][obj])
            failed = True
        else:
            lb = int(cur["lb"][v])
            stack.append((cur.copy(), v, lb))
            failed = node(cur, v, True, lb)
        while failed and stack:
            snap, v, lb = stack.pop()
            cur = snap
            failed = node(cur, v, False, lb)
        if failed and not stack:
            break
    r = np.array([x[:7] for x in rows], dtype=float)
    bb = np.array([x[8] for x in rows])
    for label, sel in (("all nodes", np.ones(len(rows), dtype=bool)), ("before the first solution", ~bb), ("with an incumbent (B&B)", bb)):
        if sel.sum() == 0:
            continue
        q = r[sel]
        print(f"{label}: {int(sel.sum())} nodes, {int(sum(1 for x, s_ in zip(rows, sel) if s_ and x[7]))} failed | unfiltered: {q[:, 0].mean():.0f} slice runs/node "
              f"({q[:, 1].mean():.0f} useful), {q[:, 2].mean():.0f} evaluations | bound-event filter: {q[:, 3].mean():.0f} runs ({q[:, 4].mean():.0f} useful), "
              f"{q[:, 5].mean():.0f} evaluations, {q[:, 6].mean():.0f} slices left entailment-dirty")



    for c, (it, rn) in sorted(iter_hist.items(), key=lambda kv: -kv[1][0]):
        print(f"   {names[c]:6s}: {rn / len(rows):7.1f} runs/node, {it / len(rows):7.1f} iterations/node, {it / max(1, rn):.2f} iterations per run")


if __name__ == "__main__":
    main()
