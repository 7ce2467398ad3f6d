"""Committed golden fixtures (tests/golden/, written by tests/golden/make_golden.py).

The reference only pins objectives (benchmarks/test_list.csv); the finer vectors were recorded from the oracle once it
reproduced those.  The CPU tests keep the oracle from drifting away from them, the GPU tests replay them on the engine
through the C-ABI without calling the oracle at all.
"""
import glob
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import BENCH, ROOT
from oracle import pyoracle
from turbo_amd import capi, frontend

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
GOLDEN = json.load(open(os.path.join(GOLDEN_DIR, "golden.json")))
NODE_FILES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "nodes_*.npz")))


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_fixtures_cover_the_reference_table():
    from conftest import known_answers
    rows = dict(known_answers())
    assert set(GOLDEN) == set(rows)
    for rel, rec in GOLDEN.items():
        assert rec["expected_objective"] == rows[rel]
    assert len(NODE_FILES) == 12


@pytest.mark.parametrize("rel", sorted(GOLDEN))
def test_oracle_reproduces_the_golden_vectors(rel):
    rec = GOLDEN[rel]
    tcn = frontend.load_fzn(os.path.join(BENCH, rel))
    assert (tcn.n_vars, tcn.n_props, tcn.n_strats) == (rec["n_vars"], rec["n_props"], rec["n_strategies"])
    root, failed, entailed, _, _ = pyoracle.propagate(tcn.store, tcn.props)
    assert (bool(failed), bool(entailed), sha(root)) == (rec["root_failed"], rec["root_all_entailed"], rec["root_fixpoint_sha256"])
    if "tree" in rec:
        has, best, st = pyoracle.solve(tcn, subproblems_power=0)
        assert has and tcn.objective_of(best) == rec["expected_objective"]
        assert {k: int(st[k]) for k in rec["tree"] if k != "best_store_sha256"} == {k: v for k, v in rec["tree"].items() if k != "best_store_sha256"}
        assert sha(best) == rec["tree"]["best_store_sha256"]


@pytest.mark.parametrize("path", NODE_FILES, ids=[os.path.basename(p)[6:-4] for p in NODE_FILES])
def test_oracle_reproduces_the_golden_nodes(path):
    z = np.load(path)
    for i in range(z["stores"].shape[0]):
        out, failed, ent, _, _ = pyoracle.propagate(z["stores"][i], z["props"])
        assert bool(failed) == bool(z["failed"][i])
        if not failed:
            assert bool(ent) == bool(z["all_entailed"][i])
            np.testing.assert_array_equal(out, z["fixpoints"][i])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["ac1", "wac1", "event", "event_compact", "wac1_removal"])
@pytest.mark.parametrize("path", NODE_FILES, ids=[os.path.basename(p)[6:-4] for p in NODE_FILES])
def test_engine_reproduces_the_golden_nodes(path, mode):
    cfg = {"ac1": dict(fixpoint=0), "wac1": dict(fixpoint=1), "event": dict(fixpoint=2), "event_compact": dict(fixpoint=2, debug=0x100000),
           "wac1_removal": dict(fixpoint=1, entailed_prop_removal=1)}[mode]
    z = np.load(path)
    got, failed, ent, _, _, _ = capi.propagate(z["props"], z["stores"], capi.make_config(**cfg))
    for i in range(z["stores"].shape[0]):
        assert bool(failed[i]) == bool(z["failed"][i]), i
        if not z["failed"][i]:
            assert bool(ent[i]) == bool(z["all_entailed"][i]), i
            np.testing.assert_array_equal(got[i], z["fixpoints"][i], err_msg=f"store {i}")


@pytest.mark.gpu
@pytest.mark.parametrize("fixpoint", [1, 2], ids=["wac1", "event"])
@pytest.mark.parametrize("rel", sorted(GOLDEN))
def test_engine_reproduces_the_golden_trees(rel, fixpoint):
    rec = GOLDEN[rel]
    tcn = frontend.load_fzn(os.path.join(BENCH, rel))
    got, failed, ent, _, _, _ = capi.propagate(tcn.props, tcn.store[None, :], capi.make_config(fixpoint=fixpoint))
    assert (bool(failed[0]), bool(ent[0]), sha(got[0])) == (rec["root_failed"], rec["root_all_entailed"], rec["root_fixpoint_sha256"])
    for key, power in (("tree", 0), ("eps6", 6)):
        if key not in rec:
            continue
        has, best, st = capi.solve(tcn, capi.make_config(or_nodes=1, subproblems_power=power, timeout_ms=120000, fixpoint=fixpoint))
        assert has and tcn.objective_of(best) == rec["expected_objective"]
        for k, v in rec[key].items():
            if k == "best_store_sha256":
                assert sha(best) == v, key
            else:
                assert int(st[k]) == v, (key, k)
