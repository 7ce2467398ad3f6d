"""One-off: random ternary networks with wide and infinite domains, node-level parity (tb_propagate vs orc_propagate).

usage: python tests/tools/fuzz_nodes.py <first seed> <last seed>
Every network has 6-24 variables and 4-40 propagators over all eight operators; a batch of perturbed stores per
network is pushed through AC1 / WAC1 / event / event+compact and compared bit for bit with the oracle.
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import pyoracle
from turbo_amd import capi

sys.path.insert(0, os.path.join(ROOT, 'tests'))
from fuzz_models import random_network, NINF, PINF


def main():
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    bad = 0
    for seed in range(lo, hi):
        rng = np.random.default_rng(seed)
        store, props = random_network(rng)
        stores = [store.copy()]
        for _ in range(7):  # perturbed copies: narrowed domains
            s = store.copy()
            for v in rng.choice(np.arange(3, s.shape[0]), size=min(3, s.shape[0] - 3), replace=False):
                l, u = int(s["lb"][v]), int(s["ub"][v])
                if l == NINF or u == PINF or l >= u:
                    continue
                m = int(rng.integers(l, u + 1))
                if rng.random() < 0.5: s["ub"][v] = m
                else: s["lb"][v] = m
            stores.append(s)
        stores = np.stack(stores)
        exp = [pyoracle.propagate(stores[i], props) for i in range(stores.shape[0])]
        for fp, dbg in ((0, 0), (1, 0), (2, 0), (2, 0x100000)):
            got, failed, ent, _, _, _ = capi.propagate(props, stores, capi.make_config(fixpoint=fp, debug=dbg, timeout_ms=20000))
            for i, (e, ef, ee, _, _) in enumerate(exp):
                ok = bool(failed[i]) == ef and (ef or (bool(ent[i]) == ee and np.array_equal(got[i], e)))
                if not ok:
                    bad += 1
                    print("MISMATCH seed", seed, "store", i, "fp", fp, hex(dbg), flush=True)
                    if bad < 4:
                        print(stores[i], props, "\nexp", e, ef, ee, "\ngot", got[i], failed[i], ent[i])
    print("seeds", lo, hi, "mismatches", bad)


main()
