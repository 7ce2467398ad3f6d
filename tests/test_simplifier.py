"""TCN simplifier (libturbo_front simplify.cpp; reference: common_solving.hpp:537-585).

The simplifier is not bit-comparable with the reference's (lala-core's simplifier is absent from the
reference tree), so it is pinned through what it must preserve: the optimum of every known-answer
instance, and that every solution of the simplified network expands to a solution of the network
as first lowered (all original propagators entailed on the expanded store).

CPU tests use the oracle as the root propagator; GPU tests use the engine (`tb_propagate`), as the
product does, and solve the simplified network on the GPU.
"""
import os

import numpy as np
import pytest

from conftest import BENCH, SLOW_FOR_ORACLE, known_answers
from oracle import pyoracle
from turbo_amd import capi, frontend, preprocess

ROWS = known_answers()
FAST = [r for r in ROWS if r[0] not in SLOW_FOR_ORACLE]
HEADLINE = ["example_wordpress7_500.fzn", "accap_a3.fzn", "trains15.fzn"]


def oracle_propagate(store, props):
    out, failed, _, _, _ = pyoracle.propagate(store, props)
    return out, failed


def check_expanded(m, tcn, best, expected):
    """`best` solves the simplified network: objective, back-mapping and validity on the original network."""
    assert tcn.objective_of(best) == expected
    o_store, o_props = m.original_network()
    full = m.expand_solution(best)
    assert full.shape[0] == o_store.shape[0]
    # inside the original root domains
    assert np.all(full["lb"] >= o_store["lb"]) and np.all(full["ub"] <= o_store["ub"])
    out, failed, entailed, _, _ = pyoracle.propagate(full, o_props)
    assert not failed, "expanded solution violates a propagator of the original network"
    assert entailed, "expanded solution leaves an original propagator undecided"
    np.testing.assert_array_equal(out, full)  # already a fixpoint: nothing left to narrow


@pytest.mark.parametrize("rel,expected", FAST)
def test_simplified_network_keeps_optimum_cpu(rel, expected):
    m, tcn, stats = preprocess.load_fzn_simplified(os.path.join(BENCH, rel), propagate=oracle_propagate)
    o_store, o_props = m.original_network()
    assert tcn.n_vars <= o_store.shape[0] and tcn.n_props <= o_props.shape[0]
    assert stats[0]["original_vars"] == o_store.shape[0] and stats[0]["original_props"] == o_props.shape[0]
    assert stats[-1]["simplified_vars"] == tcn.n_vars and stats[-1]["simplified_props"] == tcn.n_props
    has, best, st = pyoracle.solve(tcn, timeout_ms=60000)
    assert has and st["exhaustive"]
    check_expanded(m, tcn, best, expected)


@pytest.mark.parametrize("rel", [r[0] for r in ROWS] + HEADLINE)
def test_simplified_network_is_well_formed(rel):
    m, tcn, stats = preprocess.load_fzn_simplified(os.path.join(BENCH, rel), propagate=oracle_propagate)
    V = tcn.n_vars
    # the interned constants stay where the engine expects them
    for v in range(3):
        assert (int(tcn.store["lb"][v]), int(tcn.store["ub"][v])) == (v, v)
    if tcn.n_props:
        for f in ("x", "y", "z"):
            assert tcn.props[f].min() >= 0 and tcn.props[f].max() < V
        assert tcn.props["op"].min() >= 0 and tcn.props["op"].max() <= 7
        # no duplicate propagators survive
        assert np.unique(tcn.props).shape[0] == tcn.n_props
    assert 0 <= tcn.obj_var < V
    assert tcn.strat_off[0] == 0 and np.all(np.diff(tcn.strat_off) >= 0)
    used = tcn.strat_vars[: int(tcn.strat_off[-1])]
    assert used.size == 0 or (used.min() >= 0 and used.max() < V)
    # the last strategy is still the whole-store default (empty variable list)
    assert tcn.strat_off[-1] == tcn.strat_off[-2]
    # simplifying a simplified network again with its own fixpoint changes nothing more
    root, failed = oracle_propagate(tcn.store, tcn.props)
    assert not failed
    again = m.simplify(root)
    assert again["simplified_vars"] <= V and again["simplified_props"] <= tcn.n_props


def test_simplifier_reductions_are_substantial():
    """Sizes the reference reports a large drop on (ternarisation leaves many aliases and reified tautologies)."""
    m, tcn, _ = preprocess.load_fzn_simplified(os.path.join(BENCH, "trains15.fzn"), propagate=oracle_propagate)
    o_store, o_props = m.original_network()
    assert tcn.n_vars < 0.6 * o_store.shape[0] and tcn.n_props < 0.6 * o_props.shape[0]
    m, tcn, _ = preprocess.load_fzn_simplified(os.path.join(BENCH, "test_data/bug2.fzn"), propagate=oracle_propagate)
    assert tcn.n_props == 0  # solved by root propagation alone


def test_simplifier_without_fixpoint_and_unsat_root():
    m = frontend.Model.from_string(
        "var 0..5: x :: output_var;\nvar 0..5: y :: output_var;\nvar 0..5: z :: output_var;\n"
        "constraint int_eq(x, y);\nconstraint int_lin_eq([1,1],[y,z],4);\nsolve minimize z;\n")
    before = m.tcn()
    st = m.simplify(None)  # structural pass only
    after = m.tcn()
    assert st["merged_variables"] >= 1 and after.n_vars < before.n_vars
    has, best, stats = pyoracle.solve(after)
    assert has and after.objective_of(best) == 0
    text = m.format_solution(best)
    assert "x = 4;" in text and "y = 4;" in text and "z = 0;" in text

    m = frontend.Model.from_string(
        "var 0..3: x :: output_var;\nconstraint int_le(x, 1);\nconstraint int_le(3, x);\nsolve satisfy;\n")
    tcn = m.tcn()
    root, failed = oracle_propagate(tcn.store, tcn.props)
    assert failed
    m.simplify(root)
    tcn = m.tcn()
    has, _, stats = pyoracle.solve(tcn)
    assert not has and stats["exhaustive"]


@pytest.mark.gpu
@pytest.mark.parametrize("fixpoint", [1, 2], ids=["wac1", "event"])
@pytest.mark.parametrize("rel,expected", ROWS)
def test_simplified_network_keeps_optimum_gpu(rel, expected, fixpoint):
    m, tcn, stats = preprocess.load_fzn_simplified(os.path.join(BENCH, rel))  # GPU root propagation
    has, best, st = capi.solve(tcn, capi.make_config(fixpoint=fixpoint, timeout_ms=120000))
    assert has and st["exhaustive"]
    check_expanded(m, tcn, best, expected)


@pytest.mark.gpu
@pytest.mark.parametrize("rel", [r[0] for r in FAST] + HEADLINE)
def test_gpu_and_oracle_preprocessing_agree(rel):
    """Same simplified network whether the root fixpoints come from the engine or the oracle."""
    _, a, sa = preprocess.load_fzn_simplified(os.path.join(BENCH, rel))
    _, b, sb = preprocess.load_fzn_simplified(os.path.join(BENCH, rel), propagate=oracle_propagate)
    assert sa == sb
    np.testing.assert_array_equal(a.store, b.store)
    np.testing.assert_array_equal(a.props, b.props)
    np.testing.assert_array_equal(a.strat_vars[: int(a.strat_off[-1])], b.strat_vars[: int(b.strat_off[-1])])


@pytest.mark.parametrize("chunk", range(8))
def test_random_models_keep_status_optimum_and_valid_solutions(chunk):
    """Fuzz: on random small models the simplified network has the same status and optimum as the network as first
    lowered, and the lower corner of every expanded solution box satisfies the original propagators."""
    from fuzz_models import random_model
    for seed in range(chunk * 60, chunk * 60 + 60):
        text = random_model(seed)
        raw = frontend.Model.from_string(text).tcn()
        has0, best0, st0 = pyoracle.solve(raw, timeout_ms=20000)
        assert st0["exhaustive"] or (raw.goal == 0 and has0), seed  # a satisfaction search stops at its first solution
        m = frontend.Model.from_string(text)
        for _ in range(16):
            t = m.tcn()
            if t.trivially_unsat:
                break
            root, failed = oracle_propagate(t.store, t.props)
            st = m.simplify(root)
            if failed or not any(st[k] for k in ("merged_variables", "cse_merges", "entailed_props", "duplicate_props", "eliminated_variables")):
                break
        t = m.tcn()
        has1, best1, st1 = pyoracle.solve(t, timeout_ms=20000)
        assert has1 == has0, (seed, text)
        if not has0:
            continue
        if raw.goal != 0:
            assert t.objective_of(best1) == raw.objective_of(best0), (seed, text)
        o_store, o_props = m.original_network()
        full = m.expand_solution(best1)
        assert np.all(full["lb"] >= o_store["lb"]) and np.all(full["ub"] <= o_store["ub"]), (seed, text)
        corner = full.copy()
        corner["ub"] = corner["lb"]
        _, failed, entailed, _, _ = pyoracle.propagate(corner, o_props)
        assert not failed and entailed, (seed, text)
