"""Every variable order x value order of the reference (barebones_dive_and_solve.hpp:193-221 input_order / first_fail / anti_first_fail / smallest / largest,
:362-387 min / max / split / reverse_split) against the oracle on the GPU, in every store layout (anti_first_fail, largest and
indomain_reverse_split had never run on a GPU).  The variable selection decodes domains through the layout's references (COMPACT8: the base of a narrow
integer travels with the strategy entry, kernels.hpp: split / order_key), so each layout gets its own run.  Bar: one workgroup walks the oracle's tree node for
node and stops on the oracle's store.
"""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import BENCH, ROOT
from leaf_rule_models import as_tcn
from oracle import pyoracle
from turbo_amd import capi, frontend
from turbo_amd.frontend import TCN

pytestmark = pytest.mark.gpu
TURBO = os.path.join(ROOT, "turbo_amd", "bin", "turbo")
COMPACT = 0x100000
COMPACT16 = COMPACT | 0x10000000
COMPACT8 = COMPACT | 0x30000000
KEEP = 0x800000
VAR_ORDERS = ["input_order", "first_fail", "anti_first_fail", "smallest", "largest"]
VAL_ORDERS = ["min", "max", "split", "reverse_split"]
LAYOUTS = [(1, 0), (2, 0), (2, COMPACT), (2, COMPACT16), (2, COMPACT8)]
LAYOUT_IDS = ["wac1", "event", "event_compact", "event_compact16", "event_compact8"]


def walk_and_compare(tcn, power, cut, tag, **cfg):
    has_o, best_o, st_o, trace, last_o = pyoracle.solve_traced(tcn, cut, power)
    s = capi.Session(tcn, capi.make_config(or_nodes=1, subproblems_power=power, stop_after_n_nodes=cut, timeout_ms=60000, **cfg))
    plan = s.plan()
    s.start()
    while not s.poll()[1]:
        pass
    has_g, best_g, st_g = s.finish()
    last_g = s.debug_last_store(0)
    s.close()
    assert has_g == has_o, tag
    for k in ("nodes", "fails", "solutions", "depth_max", "eps_solved_subproblems", "eps_skipped_subproblems"):
        assert st_g[k] == st_o[k], (tag, k)
    if has_o:
        np.testing.assert_array_equal(best_g, best_o, err_msg=str(tag))
    if len(trace) and not trace[-1] and st_o["nodes"] == cut:
        np.testing.assert_array_equal(last_g, last_o, err_msg=str(tag))
    return plan


@pytest.mark.parametrize("val_order", range(4), ids=VAL_ORDERS)
@pytest.mark.parametrize("var_order", range(5), ids=VAR_ORDERS)
@pytest.mark.parametrize("fixpoint,debug", LAYOUTS, ids=LAYOUT_IDS)
def test_every_order_on_class_pure_networks(fixpoint, debug, var_order, val_order):
    """Fuzzed satisfiable networks (tests/fuzz_models.py: finite_class_network; scale 8 puts widths on both sides of COMPACT8's 255) under one strategy over
    the whole store with the given orders, 2^0 and 2^3 subproblems, 500 nodes."""
    from fuzz_models import finite_class_network
    used8 = 0
    for seed in range(4):
        scale = (1, 8)[seed & 1]
        rng = np.random.default_rng(31000 + 97 * var_order + 13 * val_order + seed)
        store, props = finite_class_network(rng, scale=scale)
        tcn = as_tcn(store, props, var_order=var_order, val_order=val_order)
        for power in (0, 3):
            plan = walk_and_compare(tcn, power, 500, (seed, power), fixpoint=fixpoint, debug=debug | KEEP)
            used8 += int(plan["kernel_opt"] == 4)
    if debug == COMPACT8:
        assert used8 > 0, "COMPACT8 was never planned: the case does not test what it says"


@pytest.mark.parametrize("fixpoint,debug", LAYOUTS, ids=LAYOUT_IDS)
def test_sequences_of_strategies_with_mixed_orders(fixpoint, debug):
    """seq_search: three strategies over disjoint thirds of the variables, each with its own orders (the cursor cur_strategy / next_unassigned moves from one
    to the next and is restored on backtrack, barebones:859-860), then the default strategy over the whole store."""
    from fuzz_models import finite_class_network
    for seed in range(10):
        rng = np.random.default_rng(32000 + seed)
        store, props = finite_class_network(rng, scale=(1, 8)[seed & 1])
        n = store.shape[0]
        perm = rng.permutation(np.arange(3, n))
        parts = np.array_split(perm, 3)
        vo = np.array([int(rng.integers(0, 5)) for _ in range(3)] + [1], dtype=np.int32)
        vl = np.array([int(rng.integers(0, 4)) for _ in range(3)] + [0], dtype=np.int32)
        off = np.cumsum([0] + [len(p) for p in parts] + [0]).astype(np.int32)
        tcn = TCN(store=store, props=props, strat_var_order=vo, strat_val_order=vl, strat_off=off, strat_vars=np.concatenate(parts).astype(np.int32))
        for power in (0, 4):
            walk_and_compare(tcn, power, 600, (seed, power, vo.tolist(), vl.tolist()), fixpoint=fixpoint, debug=debug | KEEP)


@pytest.mark.parametrize("rel", ["test_data/pat2.fzn", "test_data/pennies5.fzn", "test_data/sudoku_opt4.fzn", "accap_a3.fzn", "trains15.fzn"])
@pytest.mark.parametrize("var_order,val_order", [("anti_first_fail", "reverse_split"), ("largest", "max"), ("largest", "reverse_split"), ("anti_first_fail", "split"), ("smallest", "reverse_split")])
def test_eps_strategy_with_the_untested_orders_on_instances(rel, var_order, val_order):
    """`-eps_var_order / -eps_value_order` (common_solving.hpp:652-667: strategy 0 drives the dive only) with the orders no benchmark file uses: one workgroup,
    2^5 subproblems, the engine's own layout and the compact tiers forced."""
    m = frontend.Model.from_file(os.path.join(BENCH, rel))
    m.push_eps_strategy(var_order, val_order)
    tcn = m.tcn()
    assert tcn.has_eps_strategy
    for debug in (0, COMPACT8):
        walk_and_compare(tcn, 5, 1500, (rel, debug), fixpoint=2, debug=debug | KEEP)


@pytest.mark.parametrize("var_order,val_order", [("anti_first_fail", "reverse_split"), ("largest", "max"), ("smallest", "split")])
def test_cli_eps_orders_walk_the_oracles_tree(var_order, val_order):
    """The CLI flags end to end: `turbo -eps_var_order V -eps_value_order W -or 1 -sub 4 -disable_simplify` proves the same optimum with the node count of the
    oracle driven by the same strategies."""
    path = os.path.join(BENCH, "test_data", "pennies5.fzn")
    m = frontend.Model.from_file(path)
    m.push_eps_strategy(var_order, val_order)
    tcn = m.tcn()
    has_o, best_o, st_o = pyoracle.solve(tcn, subproblems_power=4)
    r = subprocess.run([TURBO, "-eps_var_order", var_order, "-eps_value_order", val_order, "-arch", "barebones", "-or", "1", "-sub", "4", "-disable_simplify", "-s", "-t", "60000", path],
                       capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr
    assert int(re.search(r"objective=(-?\d+)", r.stdout).group(1)) == tcn.objective_of(best_o) == 5
    assert int(re.search(r"mzn-stat: nodes=(\d+)", r.stdout).group(1)) == st_o["nodes"]
    assert int(re.search(r"mzn-stat: failures=(\d+)", r.stdout).group(1)) == st_o["fails"]
    assert f"-eps_var_order {var_order}" in r.stdout.splitlines()[0]
