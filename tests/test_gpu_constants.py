"""Constants kept out of the store slab (COMPACT layouts, engine.hip: Layout / operand_field, kernels.hpp: load_dom).

TCN has no constants, only singleton variables (common_solving.hpp:743-771).  The compact layouts do not give a non-Boolean singleton
of the root a slot in the workgroup's slab: the records carry its value.  These tests hit what that changes -- a rule that tries to
move a constant (failure), an objective that is a constant (pinned in the slab), strategies that list constants, batches whose
stores disagree on what is a constant -- against the oracle, and every search against the same search with the constants kept in
the slab (tb_config.reserved[0] & 0x40000000).
"""
import numpy as np
import pytest

from oracle import pyoracle
from turbo_amd import capi, frontend

pytestmark = pytest.mark.gpu

COMPACT = 0x100000
COMPACT16 = COMPACT | 0x10000000
KEEP_IN = 0x40000000
MODES = {"event_compact": dict(fixpoint=2, debug=COMPACT), "event_compact16": dict(fixpoint=2, debug=COMPACT16),
         "event_compact_globalmem": dict(fixpoint=2, debug=COMPACT, only_global_memory=1), "wac1_compact": dict(fixpoint=1, debug=COMPACT),
         "event_compact_constants_in": dict(fixpoint=2, debug=COMPACT | KEEP_IN)}


def model(text):
    return frontend.Model.from_string(text).tcn()


def check_batch(tcn, stores, **cfg):
    got, failed, ent, _, _, _ = capi.propagate(tcn.props, stores, capi.make_config(**cfg))
    for i in range(stores.shape[0]):
        exp, efailed, eent, _, _ = pyoracle.propagate(stores[i], tcn.props)
        assert bool(failed[i]) == efailed, f"store {i}: failed flag differs"
        if not efailed:
            assert bool(ent[i]) == eent, f"store {i}: entailment flag differs"
            np.testing.assert_array_equal(got[i]["lb"], exp["lb"], err_msg=f"store {i}")
            np.testing.assert_array_equal(got[i]["ub"], exp["ub"], err_msg=f"store {i}")


# a Boolean keeps the compact layouts eligible; k = 700000 does not fit 16 bits, c = 12 does
LINEAR = """var bool: b; var 0..20: x; var 0..20: y; var 700000..700000: k; var 12..12: c; var 0..800000: s;
constraint int_lin_eq([1,1,-1],[x,y,c],0); constraint int_lin_eq([1,1,-1],[x,k,s],0); constraint int_le_reif(x,5,b);
solve minimize s;"""


@pytest.mark.parametrize("mode", sorted(MODES))
def test_root_and_nodes_with_constants_bit_exact(mode):
    tcn = model(LINEAR)
    stores = [tcn.store.copy()]
    rng = np.random.default_rng(3)
    for _ in range(24):
        st = tcn.store.copy()
        for v in rng.choice(tcn.n_vars, size=3, replace=False):
            lo, hi = int(st["lb"][v]), int(st["ub"][v])
            if lo < hi:
                mid = int(rng.integers(lo, hi + 1))
                if rng.random() < 0.5:
                    st["ub"][v] = mid
                else:
                    st["lb"][v] = mid
        stores.append(st)
    check_batch(tcn, np.stack(stores), **MODES[mode])


@pytest.mark.parametrize("mode", sorted(MODES))
def test_a_rule_that_would_move_a_constant_fails_the_node(mode):
    # x + y = 12 with x, y in 0..5: the sum's candidate 0..10 empties the constant 12
    tcn = model("var bool: b; var 0..5: x; var 0..5: y; var 12..12: c; constraint int_lin_eq([1,1,-1],[x,y,c],0); constraint int_le_reif(x,2,b); solve satisfy;")
    _, failed, _, _, _, _ = capi.propagate(tcn.props, tcn.store[None, :], capi.make_config(**MODES[mode]))
    assert bool(failed[0]) and pyoracle.propagate(tcn.store, tcn.props)[1]
    has, _, st = capi.solve(tcn, capi.make_config(timeout_ms=20000, **MODES[mode]))
    assert not has and st["exhaustive"] == 1 and st["solutions"] == 0
    # ... and a comparison between two constants that is false
    tcn = model("var bool: b; var 0..3: x; var 7..7: c; var 5..5: d; constraint int_le(c,d); constraint int_le_reif(x,2,b); solve satisfy;")
    _, failed, _, _, _, _ = capi.propagate(tcn.props, tcn.store[None, :], capi.make_config(**MODES[mode]))
    assert bool(failed[0]) and pyoracle.propagate(tcn.store, tcn.props)[1]


@pytest.mark.parametrize("mode", sorted(MODES))
def test_search_tree_identical_with_constants_out(mode):
    tcn = model(LINEAR)
    has_o, best_o, st_o = pyoracle.solve(tcn, subproblems_power=0, timeout_ms=60000)
    has_g, best_g, st_g = capi.solve(tcn, capi.make_config(or_nodes=1, subproblems_power=0, timeout_ms=60000, **MODES[mode]))
    assert has_g == has_o and st_g["exhaustive"] == 1
    for k in ("nodes", "fails", "solutions", "depth_max"):
        assert st_g[k] == st_o[k], k
    np.testing.assert_array_equal(best_g, best_o)


@pytest.mark.parametrize("mode", ["event_compact", "event_compact16", "wac1_compact"])
def test_constant_objective_and_strategies_over_constants(mode):
    # the objective is a singleton of the root (pinned in the slab); the first strategy lists only constants, the second mixes them
    text = """var bool: b; var 0..9: x; var 0..9: y; var 4..4: k; var 40000..40000: big; var 4..4: obj;
constraint int_lin_eq([1,1,-1],[x,y,big],-39991); constraint int_le_reif(x,k,b); constraint int_le(k,obj);
solve :: seq_search([int_search([k,big],input_order,indomain_min,complete), int_search([big,x,k,y,b],first_fail,indomain_max,complete)]) minimize obj;"""
    tcn = model(text)
    has_o, best_o, st_o = pyoracle.solve(tcn, subproblems_power=0, timeout_ms=60000)
    has_g, best_g, st_g = capi.solve(tcn, capi.make_config(or_nodes=1, subproblems_power=0, timeout_ms=60000, **MODES[mode]))
    assert has_g and has_o and tcn.objective_of(best_g) == 4
    for k in ("nodes", "fails", "solutions", "depth_max"):
        assert st_g[k] == st_o[k], k
    np.testing.assert_array_equal(best_g, best_o)


@pytest.mark.parametrize("mode", sorted(MODES))
def test_a_batch_whose_stores_disagree_on_what_is_constant(mode):
    # x is a singleton in the first store only: it must stay a variable of the slab for the whole batch
    tcn = model("var bool: b; var 0..9: x; var 0..9: y; var 30000..30000: k; var 0..40000: s; constraint int_lin_eq([1,1,-1],[x,k,s],0); constraint int_le(y,x); constraint int_le_reif(x,3,b); solve satisfy;")
    a, b2 = tcn.store.copy(), tcn.store.copy()
    # pick a non-Boolean, non-singleton variable of the lowered network and fix it in store a only
    free = [v for v in range(tcn.n_vars) if tcn.store["lb"][v] < tcn.store["ub"][v] and tcn.store["ub"][v] > 1]
    v = free[0]
    a["lb"][v] = a["ub"][v] = int(tcn.store["lb"][v]) + 2
    b2["lb"][v] = int(tcn.store["lb"][v]) + 1
    check_batch(tcn, np.stack([a, b2, tcn.store.copy()]), **MODES[mode])


@pytest.mark.parametrize("rel", ["example_wordpress7_500.fzn", "trains15.fzn"])
def test_headline_search_is_the_same_with_the_constants_in_or_out(rel):
    """Full grid, node budget: the tree does not depend on the layout -- same counters after a deterministic one-workgroup prefix."""
    import os
    from conftest import BENCH
    from turbo_amd import preprocess
    tcn = preprocess.load_fzn_simplified(os.path.join(BENCH, rel))[1]
    res = []
    for dbg in (0, KEEP_IN):
        has, best, st = capi.solve(tcn, capi.make_config(or_nodes=1, subproblems_power=4, stop_after_n_nodes=1500, timeout_ms=120000, fixpoint=2, debug=dbg))
        res.append((has, None if not has else best.tobytes(), st["nodes"], st["fails"], st["solutions"], st["depth_max"], st["eps_solved_subproblems"], st["eps_skipped_subproblems"]))
    assert res[0] == res[1]
