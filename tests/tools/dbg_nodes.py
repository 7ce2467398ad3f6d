import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from turbo_amd import capi, frontend
from oracle import pyoracle
from test_gpu_parity import random_nodes
tcn = frontend.load_fzn(os.path.join(ROOT, "benchmarks", "example_wordpress7_500.fzn"))
stores = random_nodes(tcn, 60, seed=3, max_decisions=60)
for bits in (0, 0x4000000):
    got, failed, ent, _, _, _ = capi.propagate(tcn.props, stores, capi.make_config(fixpoint=2, debug=bits))
    bad = 0
    for i in range(stores.shape[0]):
        exp, efailed, eent, _, _ = pyoracle.propagate(stores[i], tcn.props)
        if bool(failed[i]) != efailed:
            print(hex(bits), "store", i, "failed flag", bool(failed[i]), efailed); bad += 1; continue
        if efailed: continue
        if bool(ent[i]) != eent: print(hex(bits), "store", i, "entailed", bool(ent[i]), eent); bad += 1
        d = np.flatnonzero((got[i]["lb"] != exp["lb"]) | (got[i]["ub"] != exp["ub"]))
        if d.size:
            bad += 1
            v = int(d[0])
            print(hex(bits), "store", i, "differs at", d.size, "vars; first", v, "got", got[i][v], "exp", exp[v], "in", stores[i][v])
            rd = [(int(p["op"]), int(p["x"]), int(p["y"]), int(p["z"])) for p in tcn.props if v in (int(p["x"]), int(p["y"]), int(p["z"]))][:12]
            print("   readers:", rd)
            for (_, x, y, z) in rd[:6]:
                print("     ", [(int(u), tuple(got[i][u]), tuple(exp[u])) for u in (x, y, z)])
            break
    print(hex(bits), "mismatching stores:", bad, "of", stores.shape[0], flush=True)
