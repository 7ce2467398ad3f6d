"""CPU tests: the oracle against the reference's own known-answer table (benchmarks/test_list.csv,
the contract of test_turbo.sh:34-67).  This is what pins the oracle end to end."""
import os

import numpy as np
import pytest

from conftest import BENCH, SLOW_FOR_ORACLE, known_answers
from oracle import pyoracle
from turbo_amd import frontend

ROWS = known_answers()


@pytest.mark.parametrize("rel,expected", ROWS)
def test_known_objective(rel, expected):
    tcn = frontend.load_fzn(os.path.join(BENCH, rel))
    slow = rel in SLOW_FOR_ORACLE
    has, best, st = pyoracle.solve(tcn, timeout_ms=6000 if slow else 120000)
    assert has
    assert tcn.objective_of(best) == expected
    if not slow:
        assert st["exhaustive"] == 1  # optimality proved
    # the recorded solution satisfies every propagator (all entailed on the box)
    _, failed, ent, _, _ = pyoracle.propagate(best, tcn.props)
    assert not failed and ent


@pytest.mark.parametrize("rel,expected", [r for r in ROWS if r[0] not in SLOW_FOR_ORACLE][:24])
@pytest.mark.parametrize("power", [0, 5])
def test_eps_order_same_optimum(rel, expected, power):
    """Sequential dive-and-solve over 2^d subproblems reaches the same optimum as the plain DFS,
    and the canonical (fixed-bound) pass returns an optimal solution."""
    tcn = frontend.load_fzn(os.path.join(BENCH, rel))
    has, best, st = pyoracle.solve(tcn, subproblems_power=power, timeout_ms=120000)
    assert has and tcn.objective_of(best) == expected and st["exhaustive"] == 1
    has2, best2, st2 = pyoracle.solve(tcn, subproblems_power=power, fixed_bound=st["best_bound"], timeout_ms=120000)
    assert has2 and tcn.objective_of(best2) == expected


def test_unsat_fixture():
    tcn = frontend.load_fzn(os.path.join(BENCH, "unsolved_bugs_data", "false.fzn"))
    assert tcn.trivially_unsat


def test_unsat_by_search():
    tcn = frontend.Model.from_string("var 1..3: x; var 1..3: y; constraint int_lt(x,y); constraint int_lt(y,x); solve satisfy;").tcn()
    has, _, st = pyoracle.solve(tcn)
    assert not has and st["exhaustive"] == 1


@pytest.mark.parametrize("rel,power,cut", [("test_data/sudoku_opt4.fzn", 0, 30), ("test_data/sudoku_opt4.fzn", 4, 77), ("test_data/pat2.fzn", 4, 55),
                                           ("accap_a3.fzn", 6, 400), ("accap_a3.fzn", 0, 37), ("test_data/pennies5.fzn", 4, 70), ("test_data/pennies5.fzn", 0, 41),
                                           ("example_wordpress7_500.fzn", 6, 30), ("test_data/triangular9.fzn", 2, 1200), ("test_data/triangular9.fzn", 0, 333)]
                         + [("test_data/triangular9.fzn", 3, c) for c in range(40, 60)] + [("accap_a3.fzn", 4, c) for c in range(100, 130)])
def test_replay_of_the_oracles_own_path_gives_its_last_store(rel, power, cut):
    """orc_replay_path (the checker of the full-grid GPU parity test) against orc_solve itself: stop the search after `cut` nodes, take the
    path it stands on (subproblem, decisions with the objective bound in force at each), replay it from the root -- the store must be
    the one the search stopped on, whatever snapshots and older bounds it had gone through."""
    from conftest import BENCH
    from turbo_amd import frontend
    tcn = frontend.load_fzn(os.path.join(BENCH, rel))
    has, best, st, hdr, dec, last, last_failed = pyoracle.solve_with_path(tcn, cut, power)
    assert st["nodes"] == cut, "the budget must cut the search for the case to mean something"
    store, failed, mismatch = pyoracle.replay_path(tcn, power, hdr, dec)
    assert mismatch == -1, (hdr, mismatch)
    assert failed == last_failed
    if not failed:
        np.testing.assert_array_equal(store, last)
