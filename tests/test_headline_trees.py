"""Tree-level parity at BASELINE size: the SEARCH of wordpress7_500, accap_a3 and trains15 -- raw
and simplified networks -- against the committed oracle vectors of tests/golden/headline_trees.json
(written by tests/golden/make_headline_golden.py).

One workgroup, 2^0 and 2^6 subproblems, a node budget: after the same number of nodes the engine must show the oracle's
node / fail / solution / depth / subproblem counters, the same best store and the same store under the last node,
bit for bit -- in the sweeping (WAC1) and in the event-driven fixpoint, with and without snapshots.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import BENCH, ROOT
from oracle import pyoracle
from turbo_amd import capi, frontend, preprocess

GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "headline_trees.json")))
COUNTERS = ("nodes", "fails", "solutions", "depth_max", "eps_solved_subproblems", "eps_skipped_subproblems")
KEEP_LAST = 0x800000  # tb_config.reserved[0]: keep the store every workgroup stopped on
# (threads per workgroup, event kernel, store layout, memory kind) of bench.py's sessions on an MI355X -- the grid differs (one workgroup here), the
# kernel instantiation must not (tests/test_gpu_fullgrid_paths.py runs the full grids)
BENCH_PLANS = {"example_wordpress7_500.fzn/simplified": (128, 1, 1, 1), "accap_a3.fzn/simplified": (128, 1, 0, 1), "trains15.fzn/simplified": (128, 1, 4, 1)}


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def network(key, propagate=None):
    name, kind = key.split("/")
    path = os.path.join(BENCH, name)
    if kind == "raw":
        return frontend.load_fzn(path)
    return preprocess.load_fzn_simplified(path, propagate=propagate)[1]


def oracle_propagate(store, props):
    out, failed, _, _, _ = pyoracle.propagate(store, props)
    return out, failed


def check_network(tcn, net):
    assert (tcn.n_vars, tcn.n_props) == (net["n_vars"], net["n_props"])
    assert sha(np.ascontiguousarray(tcn.props)) + sha(np.ascontiguousarray(tcn.store)) == net["network_sha256"]


def test_fixture_covers_the_three_headline_instances():
    assert sorted(GOLDEN) == sorted(f"{n}/{k}" for n in ("example_wordpress7_500.fzn", "accap_a3.fzn", "trains15.fzn") for k in ("raw", "simplified"))
    for net in GOLDEN.values():
        assert sorted(net["cases"]) == ["sub0_cut2000", "sub0_cut500", "sub6_cut2000", "sub6_cut500"]


@pytest.mark.parametrize("key", sorted(GOLDEN))
def test_oracle_reproduces_the_headline_vectors(key):
    """CPU: the oracle still walks the recorded tree (the 500-node cases: a few seconds per network)."""
    tcn = network(key, propagate=oracle_propagate)
    check_network(tcn, GOLDEN[key])
    for case, rec in GOLDEN[key]["cases"].items():
        if rec["cutnodes"] > 500:
            continue
        has, best, st, trace, last = pyoracle.solve_traced(tcn, rec["cutnodes"], rec["subproblems_power"])
        for k in COUNTERS + ("best_bound",):
            assert int(st[k]) == rec[k], (case, k)
        assert (sha(best) if has else None) == rec["best_store_sha256"], case
        assert bool(trace[-1]) == rec["last_node_failed"]
        if not rec["last_node_failed"]:
            assert sha(last) == rec["last_store_sha256"], case


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["wac1", "event", "event_recompute", "wac1_rm", "event_padded", "event_unpadded", "event_no_compact8"])
@pytest.mark.parametrize("key", sorted(GOLDEN))
def test_engine_reproduces_the_headline_trees(key, mode):
    """GPU: the same prefix of the tree through the C-ABI, without running the oracle.
    (event_padded / event_unpadded: every class of records padded to whole slices, or never -- the engine pads when the store stays in
    LDS with the padded records; test knobs 0x10 / 0x20 of tb_config.reserved[0].  event_no_compact8: sign bit, the two-byte integer tier
    never taken -- trains15 then runs on COMPACT16 slabs as in r03.)"""
    cfg = {"wac1": dict(fixpoint=1), "event": dict(fixpoint=2), "event_recompute": dict(fixpoint=2, snapshot_levels=1),
           "wac1_rm": dict(fixpoint=1, entailed_prop_removal=1), "event_padded": dict(fixpoint=2, debug_extra=0x10),
           "event_unpadded": dict(fixpoint=2, debug_extra=0x20), "event_no_compact8": dict(fixpoint=2, debug_extra=-0x80000000)}[mode]
    extra = cfg.pop("debug_extra", 0)
    tcn = network(key)  # simplified: root fixpoints by the engine itself (tb_propagate) -- the network must come out identical
    check_network(tcn, GOLDEN[key])
    for case, rec in GOLDEN[key]["cases"].items():
        s = capi.Session(tcn, capi.make_config(or_nodes=1, subproblems_power=rec["subproblems_power"], stop_after_n_nodes=rec["cutnodes"],
                                               timeout_ms=120000, debug=KEEP_LAST | extra, **cfg))
        if mode == "event" and key in BENCH_PLANS:  # the kernel instantiation and store placement are the ones the bench line reports
            plan = s.plan()
            info = capi.device_info(0)
            if info["compute_units"] == 256 and info["lds_bytes_per_cu"] == 160 * 1024:
                assert (plan["threads_per_block"], plan["kernel_event"], plan["kernel_opt"], plan["mem_kind"]) == BENCH_PLANS[key], (key, plan)
        s.start()
        while not s.poll()[1]:
            pass
        has, best, st = s.finish()
        last = s.debug_last_store(0)
        s.close()
        for k in COUNTERS:
            assert int(st[k]) == rec[k], (case, k)
        assert (st["best_bound"] if has else capi.TB_PINF) == rec["best_bound"], case
        assert (sha(best) if has else None) == rec["best_store_sha256"], case
        if not rec["last_node_failed"]:
            assert sha(last) == rec["last_store_sha256"], case
