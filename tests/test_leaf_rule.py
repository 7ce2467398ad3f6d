"""CPU tests of the oracle's two leaf rules (orc_config.leaf_requires_assignment).

The reference has two: barebones calls a node a solution as soon as no propagator is active (barebones_dive_and_solve.hpp:988-993) -- the solution is a box
whose open variables are free --, the gpu and cpu paths additionally ask that the store be extractable, i.e. that every variable be assigned
(gpu_dive_and_solve.hpp:333-338, cpu_solving.hpp:33-40, hybrid_dive_and_solve.hpp:531), and keep branching otherwise.  Pinned here by
  * the reference-held objectives of benchmarks/test_list.csv under the gpu / cpu rule too (same optimum, a fully assigned best store);
  * brute force, independent of oracle/ (tests/rule_brute.py): under the gpu / cpu rule the solutions of a satisfaction problem are exactly the full assignments
    that satisfy every propagator, each once; under barebones' rule the solution boxes are disjoint, every point of a box is such an assignment and
    their volumes add up to the same number;
  * the path replay that checks the full-grid GPU runs (orc_replay_path), against the search itself, under the rule.
"""
import os

import numpy as np
import pytest

from conftest import BENCH, SLOW_FOR_ORACLE, known_answers
from leaf_rule_models import as_tcn, box_volume, brute_force_solutions, loose_network
from oracle import pyoracle
from turbo_amd import frontend

ROWS = known_answers()
FAST = [r for r in ROWS if r[0] not in SLOW_FOR_ORACLE]


@pytest.mark.parametrize("rel,expected", FAST)
def test_reference_objective_under_the_gpu_leaf_rule(rel, expected):
    tcn = frontend.load_fzn(os.path.join(BENCH, rel))
    has, best, st = pyoracle.solve(tcn, timeout_ms=120000, leaf_requires_assignment=1)
    assert has and st["exhaustive"] == 1
    assert tcn.objective_of(best) == expected
    assert (best["lb"] == best["ub"]).all(), "is_extractable: every variable of a solution is assigned"
    _, failed, ent, _, _ = pyoracle.propagate(best, tcn.props)
    assert not failed and ent
    # barebones' rule stops at the box, so it can only visit fewer (or the same) nodes on the way to the same optimum
    has0, best0, st0 = pyoracle.solve(tcn, timeout_ms=120000)
    assert tcn.objective_of(best0) == expected and st0["nodes"] <= st["nodes"]


@pytest.mark.parametrize("chunk", range(4))
def test_solutions_are_exactly_the_brute_force_assignments(chunk):
    differ = 0
    for seed in range(100 * chunk, 100 * chunk + 25):
        rng = np.random.default_rng(4200 + seed)
        store, props = loose_network(rng)
        tcn = as_tcn(store, props, var_order=int(rng.integers(0, 5)), val_order=int(rng.integers(0, 4)))
        expected = brute_force_solutions(store, props)
        # gpu / cpu rule: full assignments, each exactly once
        sols, st = pyoracle.enumerate_solutions(tcn, leaf_requires_assignment=1)
        assert st["exhaustive"] == 1 and st["solutions"] == len(sols) == len(expected), seed
        got = {tuple(int(x) for x in s["lb"]) for s in sols}
        assert all((s["lb"] == s["ub"]).all() for s in sols)
        assert got == set(expected), seed
        # barebones' rule: disjoint boxes covering the same set
        boxes, st0 = pyoracle.enumerate_solutions(tcn, leaf_requires_assignment=0)
        assert st0["exhaustive"] == 1
        assert sum(box_volume(b) for b in boxes) == len(expected), seed
        for b in boxes:
            assert all(all(int(b["lb"][v]) <= a[v] <= int(b["ub"][v]) for v in range(len(a))) for a in brute_force_solutions(b, props))
            assert len(brute_force_solutions(b, props)) == box_volume(b), "every point of a solution box satisfies every propagator"
        differ += int(len(boxes) != len(sols))
        assert st0["nodes"] <= st["nodes"]
    assert differ >= 10, "the family is meant to tell the two rules apart"


def test_a_variable_no_propagator_fixes():
    """x < y over 1..3 and a variable z in 0..2 nothing mentions: 3 (x, y) pairs, 9 assignments.  barebones reports two boxes (x = 1, y in 2..3, z open; x = 2, y = 3,
    z open), the gpu / cpu rule the 9 assignments."""
    tcn = frontend.Model.from_string("var 1..3: x :: output_var; var 1..3: y :: output_var; var 0..2: z :: output_var; constraint int_lt(x, y); solve satisfy;").tcn()
    boxes, st0 = pyoracle.enumerate_solutions(tcn)
    sols, st1 = pyoracle.enumerate_solutions(tcn, leaf_requires_assignment=1)
    assert st0["solutions"] == 2 and st1["solutions"] == 9
    assert sum(box_volume(b) for b in boxes) == 9
    assert all((s["lb"] == s["ub"]).all() for s in sols)
    assert any((b["lb"] != b["ub"]).any() for b in boxes)
    assert len(brute_force_solutions(tcn.store, tcn.props)) == 9


@pytest.mark.parametrize("rel,power,cut", [("test_data/sudoku_opt4.fzn", 4, 77), ("accap_a3.fzn", 6, 400), ("test_data/pennies5.fzn", 0, 41), ("test_data/pat2.fzn", 4, 55)]
                         + [("test_data/pennies5.fzn", 3, c) for c in range(60, 75)])
def test_replay_of_the_oracles_own_path_under_the_gpu_rule(rel, power, cut):
    tcn = frontend.load_fzn(os.path.join(BENCH, rel))
    has, best, st, hdr, dec, last, last_failed = pyoracle.solve_with_path(tcn, cut, power, leaf_requires_assignment=1)
    assert st["nodes"] == cut
    store, failed, mismatch = pyoracle.replay_path(tcn, power, hdr, dec, leaf_requires_assignment=1)
    assert mismatch == -1, (hdr, mismatch)
    assert failed == last_failed
    if not failed:
        np.testing.assert_array_equal(store, last)


def test_replay_through_all_entailed_inner_nodes():
    """Paths that pass THROUGH all-entailed nodes with open variables (inner nodes under the gpu / cpu rule, leaves under barebones'): the replay must follow the rule."""
    through = 0
    for seed in range(40):
        rng = np.random.default_rng(777 + seed)
        store, props = loose_network(rng, n_free=2)
        tcn = as_tcn(store, props)
        for cut in (3, 5, 8, 13):
            has, best, st, hdr, dec, last, last_failed = pyoracle.solve_with_path(tcn, cut, 0, leaf_requires_assignment=1)
            if st["nodes"] < cut:
                continue
            store1, failed, mismatch = pyoracle.replay_path(tcn, 0, hdr, dec, leaf_requires_assignment=1)
            assert mismatch == -1, (seed, cut, hdr)
            assert failed == last_failed
            if not failed:
                np.testing.assert_array_equal(store1, last)
            # the same path replayed under barebones' rule stops at the first all-entailed node, if the path crosses one
            _, _, mm0 = pyoracle.replay_path(tcn, 0, hdr, dec, leaf_requires_assignment=0)
            through += int(mm0 != -1)
    assert through > 0
