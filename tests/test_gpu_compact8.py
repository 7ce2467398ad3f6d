"""COMPACT8 store layout of the event kernels (kernels.hpp: layout 4; engine.hip: Layout::c8): an integer whose root domain is at most 255 wide is
two bytes of the slab -- its bounds relative to the root lower bound, which travels with every reference to the variable -- the wider ones are 16-bit
pairs, the Booleans 2 bits.  Same fixpoints, same trees, bit for bit: batches of nodes against the oracle on networks whose domains straddle every
limit of the packing (width 255, bases beyond +-16383, values beyond 16 bits), one workgroup walking the oracle's tree with the tier forced, and the
engine's own choice on trains15 (the instance the tier was built for: eleven or twelve two-wave workgroups per CU instead of seven four-wave ones)."""
import os

import numpy as np
import pytest

from conftest import BENCH
from oracle import pyoracle
from turbo_amd import capi, frontend, preprocess
from turbo_amd.frontend import TCN

pytestmark = pytest.mark.gpu
COMPACT8 = 0x100000 | 0x30000000  # tb_config.reserved[0]: compact layout forced, two-byte tier whenever the network is eligible
KEEP = 0x800000


def as_tcn(store, props):
    """One input-order / min-value strategy over all variables, satisfaction."""
    z = np.zeros(1, dtype=np.int32)
    return TCN(store=store, props=props, strat_var_order=z, strat_val_order=z, strat_off=np.zeros(2, dtype=np.int32), strat_vars=np.zeros(0, dtype=np.int32))


def eligible(store):
    """What engine.hip: make_layout demands of a root store."""
    lb, ub = store["lb"].astype(np.int64), store["ub"].astype(np.int64)
    boolean = (lb >= 0) & (ub <= 1)
    ints = ~boolean & (lb != ub)  # (the constants leave the slab)
    narrow = ints & (ub - lb <= 255) & (lb >= -16383) & (lb <= 16382)
    return bool(boolean.any() and narrow.any() and (lb[ints] >= -32768).all() and (ub[ints] <= 32767).all() and len(lb) < 0xffff)


@pytest.mark.parametrize("scale", [1, 8, 400, 1500])
def test_nodes_of_networks_straddling_the_packing_limits(scale):
    from fuzz_models import finite_class_network
    taken = 0
    for seed in range(40):
        rng = np.random.default_rng(9000 + 100 * scale + seed)
        store, props = finite_class_network(rng, scale=scale)
        root, failed, _, _, _ = pyoracle.propagate(store, props)
        stores = [store]
        for _ in range(7):
            s = (store if failed or rng.random() < 0.3 else root).copy()
            for v in rng.choice(np.arange(3, s.shape[0]), size=min(int(rng.integers(1, 8)), s.shape[0] - 3), replace=False):
                lo, hi = int(s["lb"][v]), int(s["ub"][v])
                if lo >= hi:
                    continue
                m = int(rng.integers(lo, hi + 1))
                if rng.random() < 0.5:
                    s["lb"][v] = m
                else:
                    s["ub"][v] = m
            stores.append(s)
        stores = np.stack(stores)
        got, gfailed, ent, _, _, _ = capi.propagate(props, stores, capi.make_config(fixpoint=2, debug=COMPACT8, timeout_ms=20000))
        for i in range(stores.shape[0]):
            exp, efailed, eent, _, _ = pyoracle.propagate(stores[i], props)
            assert bool(gfailed[i]) == efailed, (seed, i)
            if not efailed:
                assert bool(ent[i]) == eent, (seed, i)
                np.testing.assert_array_equal(got[i], exp, err_msg=f"scale {scale} seed {seed} store {i}")
        if seed % 8 == 0:  # the plan of a search on the same network: is the tier what ran?
            s = capi.Session(as_tcn(store, props), capi.make_config(fixpoint=2, debug=COMPACT8, or_nodes=1, subproblems_power=0, timeout_ms=20000))
            plan = s.plan()
            s.close()
            assert (plan["kernel_opt"] == 4) == eligible(store), (scale, seed, plan)
            taken += plan["kernel_opt"] == 4
    assert taken >= (3 if scale <= 400 else 0), "the fuzz must exercise the tier"


@pytest.mark.parametrize("scale", [1, 8, 400])
@pytest.mark.parametrize("threads", [0, 256])
def test_one_workgroup_walks_the_oracles_tree_on_fuzzed_networks(scale, threads):
    from fuzz_models import finite_class_network
    for seed in range(10):
        rng = np.random.default_rng(9500 + 100 * scale + seed)
        store, props = finite_class_network(rng, scale=scale)
        tcn = as_tcn(store, props)
        for power in (0, 3):
            has_o, best_o, st_o, trace, last_o = pyoracle.solve_traced(tcn, 400, power)
            s = capi.Session(tcn, capi.make_config(fixpoint=2, debug=COMPACT8 | KEEP, or_nodes=1, subproblems_power=power, stop_after_n_nodes=400, threads_per_block=threads, timeout_ms=60000))
            plan = s.plan()
            s.start()
            while not s.poll()[1]:
                pass
            has_g, best_g, st_g = s.finish()
            last_g = s.debug_last_store(0)
            s.close()
            assert (plan["kernel_opt"] == 4) == eligible(store), plan
            assert has_g == has_o, (seed, power)
            for k in ("nodes", "fails", "solutions", "depth_max", "eps_solved_subproblems", "eps_skipped_subproblems"):
                assert st_g[k] == st_o[k], (seed, power, k)
            if has_o:
                np.testing.assert_array_equal(best_g, best_o)
            if not trace[-1]:
                np.testing.assert_array_equal(last_g, last_o, err_msg=f"seed {seed} power {power}")


@pytest.mark.parametrize("name,cut", [("trains15.fzn", 3000), ("accap_a3.fzn", 3000), ("test_data/pat7.fzn", 2000), ("test_data/sudoku_opt4.fzn", 2000)])
@pytest.mark.parametrize("power", [0, 5])
def test_one_workgroup_walks_the_oracles_tree_on_instances(name, cut, power):
    path = os.path.join(BENCH, name)
    tcn = preprocess.load_fzn_simplified(path)[1] if "/" not in name else frontend.load_fzn(path)
    has_o, best_o, st_o, trace, last_o = pyoracle.solve_traced(tcn, cut, power)
    s = capi.Session(tcn, capi.make_config(fixpoint=2, debug=COMPACT8 | KEEP, or_nodes=1, subproblems_power=power, stop_after_n_nodes=cut, timeout_ms=240000))
    plan = s.plan()
    s.start()
    while not s.poll()[1]:
        pass
    has_g, best_g, st_g = s.finish()
    last_g = s.debug_last_store(0)
    s.close()
    assert plan["kernel_opt"] == 4 and plan["mem_kind"] == 1, plan
    assert has_g == has_o
    for k in ("nodes", "fails", "solutions", "depth_max", "eps_solved_subproblems", "eps_skipped_subproblems"):
        assert st_g[k] == st_o[k], k
    if has_o:
        np.testing.assert_array_equal(best_g, best_o)
    if not trace[-1]:
        np.testing.assert_array_equal(last_g, last_o)


def test_the_engine_takes_the_tier_for_trains15_and_not_for_the_headline():
    info = capi.device_info(0)
    if not (info["compute_units"] == 256 and info["lds_bytes_per_cu"] == 160 * 1024):
        pytest.skip("plans of an MI355X")
    for name, want in (("trains15.fzn", dict(kernel_opt=4, threads_per_block=128, mem_kind=1)), ("example_wordpress7_500.fzn", dict(kernel_opt=1, threads_per_block=128, mem_kind=1)),
                       ("accap_a3.fzn", dict(kernel_opt=0, threads_per_block=128, mem_kind=1))):
        _, tcn, _ = preprocess.load_fzn_simplified(os.path.join(BENCH, name))
        for bits, opt in ((0, want["kernel_opt"]), (0x80000000 - (1 << 32), None)):  # sign bit: never the tier
            s = capi.Session(tcn, capi.make_config(fixpoint=2, debug=bits, timeout_ms=60000))
            plan = s.plan()
            s.close()
            if opt is not None:
                for k, v in want.items():
                    assert plan[k] == v, (name, plan)
            else:
                assert plan["kernel_opt"] != 4, (name, plan)
    _, tcn, _ = preprocess.load_fzn_simplified(os.path.join(BENCH, "trains15.fzn"))
    s = capi.Session(tcn, capi.make_config(fixpoint=2, timeout_ms=60000))
    assert s.plan()["num_blocks"] >= 11 * 256
    s.close()
