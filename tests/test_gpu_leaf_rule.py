"""GPU tests of the two leaf rules (tb_config.leaf_requires_assignment; oracle twin: tests/test_leaf_rule.py).

`-arch barebones`: a node whose propagators are all entailed is a solution (barebones_dive_and_solve.hpp:988-993).  `-arch gpu` (and the reference's cpu path):
... only if the store is extractable as well, i.e. every variable is assigned (gpu_dive_and_solve.hpp:333-338, cpu_solving.hpp:33-40); an all-entailed node with
an open variable is an inner node and the search keeps branching below it.  Same optimum, different trees, different printed solutions and -- under `-a` -- a
different NUMBER of solutions whenever some variable is fixed by no propagator.  Bar: one workgroup walks the oracle's tree node for node under either rule;
the solutions streamed under the gpu rule are exactly the brute-force full assignments, each once.
"""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import BENCH, ROOT, SLOW_FOR_ORACLE, known_answers
from leaf_rule_models import as_tcn, box_volume, brute_force_solutions, loose_network
from oracle import pyoracle
from test_gpu_streaming import run_streaming
from turbo_amd import capi, frontend, preprocess

pytestmark = pytest.mark.gpu
TURBO = os.path.join(ROOT, "turbo_amd", "bin", "turbo")
ROWS = known_answers()
FAST = [r for r in ROWS if r[0] not in SLOW_FOR_ORACLE]
COMPACT = 0x100000
COMPACT16 = COMPACT | 0x10000000
COMPACT8 = COMPACT | 0x30000000
KEEP = 0x800000
MODES = [(1, 0), (2, 0), (2, COMPACT), (1, "rm"), (2, COMPACT16), (2, COMPACT8), (0, 0)]
MODE_IDS = ["wac1", "event", "event_compact", "wac1_rm", "event_compact16", "event_compact8", "ac1"]


@pytest.mark.parametrize("fixpoint,debug", MODES, ids=MODE_IDS)
@pytest.mark.parametrize("rel,expected", FAST)
def test_sequential_tree_identical_under_the_gpu_rule(rel, expected, fixpoint, debug):
    """One workgroup, one subproblem: the oracle's DFS tree under leaf_requires_assignment = 1, and a fully assigned best store."""
    rm, debug = (1, 0) if debug == "rm" else (0, debug)
    tcn = frontend.load_fzn(os.path.join(BENCH, rel))
    has_o, best_o, st_o = pyoracle.solve(tcn, subproblems_power=0, timeout_ms=120000, leaf_requires_assignment=1)
    has_g, best_g, st_g = capi.solve(tcn, capi.make_config(or_nodes=1, subproblems_power=0, timeout_ms=120000, fixpoint=fixpoint, debug=debug, entailed_prop_removal=rm,
                                                           leaf_requires_assignment=1))
    assert has_g == has_o and st_g["exhaustive"] == st_o["exhaustive"] == 1
    assert tcn.objective_of(best_g) == expected
    for k in ("nodes", "fails", "solutions", "depth_max"):
        assert st_g[k] == st_o[k], k
    np.testing.assert_array_equal(best_g, best_o)
    assert (best_g["lb"] == best_g["ub"]).all()


@pytest.mark.parametrize("rel", ["test_data/sudoku_opt4.fzn", "test_data/pat2.fzn", "test_data/pat7.fzn", "test_data/pennies5.fzn"])
@pytest.mark.parametrize("power", [3, 6])
@pytest.mark.parametrize("fixpoint,levels,debug", [(1, 0, 0), (2, 0, 0), (2, 1, 0), (2, 0, COMPACT), (2, 1, COMPACT8), (1, 1, "rm")],
                         ids=["wac1", "event", "event_recompute", "event_compact", "event_compact8_recompute", "wac1_rm_recompute"])
def test_sequential_eps_identical_under_the_gpu_rule(rel, power, fixpoint, levels, debug):
    rm, debug = (1, 0) if debug == "rm" else (0, debug)
    tcn = frontend.load_fzn(os.path.join(BENCH, rel))
    has_o, best_o, st_o = pyoracle.solve(tcn, subproblems_power=power, leaf_requires_assignment=1)
    has_g, best_g, st_g = capi.solve(tcn, capi.make_config(or_nodes=1, subproblems_power=power, timeout_ms=120000, fixpoint=fixpoint, snapshot_levels=levels, debug=debug,
                                                           entailed_prop_removal=rm, leaf_requires_assignment=1))
    assert has_g == has_o
    for k in ("nodes", "fails", "solutions", "depth_max", "eps_solved_subproblems", "eps_skipped_subproblems"):
        assert st_g[k] == st_o[k], k
    np.testing.assert_array_equal(best_g, best_o)


@pytest.mark.parametrize("chunk", range(3))
@pytest.mark.parametrize("fixpoint,debug", [(1, 0), (2, 0), (2, COMPACT), (2, COMPACT8)], ids=["wac1", "event", "event_compact", "event_compact8"])
def test_loose_networks_both_rules_walk_the_oracles_tree(chunk, fixpoint, debug):
    """Networks with variables no propagator fixes (tests/leaf_rule_models.py): under either rule one workgroup visits the oracle's nodes, in 2^0 and 2^3
    subproblems, with every variable / value order of the reference; the two rules must differ on most of them."""
    differ = 0
    for seed in range(100 * chunk, 100 * chunk + 20):
        rng = np.random.default_rng(4200 + seed)
        store, props = loose_network(rng)
        tcn = as_tcn(store, props, var_order=int(rng.integers(0, 5)), val_order=int(rng.integers(0, 4)))
        counts = []
        for rule in (0, 1):
            for power in (0, 3):
                has_o, best_o, st_o = pyoracle.solve(tcn, subproblems_power=power, stop_after_n_solutions=0, leaf_requires_assignment=rule)
                has_g, best_g, st_g = capi.solve(tcn, capi.make_config(or_nodes=1, subproblems_power=power, timeout_ms=60000, fixpoint=fixpoint, debug=debug,
                                                                       stop_after_n_solutions=0, leaf_requires_assignment=rule))
                assert has_g == has_o and st_g["exhaustive"] == st_o["exhaustive"], (seed, rule, power)
                for k in ("nodes", "fails", "solutions", "depth_max", "eps_solved_subproblems", "eps_skipped_subproblems"):
                    assert st_g[k] == st_o[k], (seed, rule, power, k)
            counts.append(st_o["solutions"])
        differ += int(counts[0] != counts[1])
    assert differ >= 8


@pytest.mark.parametrize("fixpoint,debug,grid", [(1, 0, 0), (2, 0, 0), (2, COMPACT, 0), (2, COMPACT8, 0), (2, 0, 1), (1, 0, 1)],
                         ids=["wac1", "event", "event_compact", "event_compact8", "event_one_workgroup", "wac1_one_workgroup"])
def test_all_solutions_are_the_brute_force_assignments(fixpoint, debug, grid):
    """`-a` under the gpu rule: the solutions handed to the host are exactly the full assignments that satisfy every propagator (brute force, independent of
    the oracle), each once -- on the whole grid (2^d subproblems racing) and on one workgroup; under barebones' rule they are disjoint boxes of the same total volume."""
    for seed in range(12):
        rng = np.random.default_rng(5100 + seed)
        store, props = loose_network(rng)
        tcn = as_tcn(store, props, var_order=int(rng.integers(0, 5)), val_order=int(rng.integers(0, 4)))
        expected = set(brute_force_solutions(store, props))
        extra = dict(or_nodes=1, subproblems_power=2) if grid else {}
        got, has, best, st = run_streaming(tcn, fixpoint=fixpoint, debug=debug, stop_after_n_solutions=0, leaf_requires_assignment=1, **extra)
        assert st["exhaustive"] and st["solutions"] == len(got) == len(expected), seed
        assert all((s["lb"] == s["ub"]).all() for s, _ in got)
        assert {tuple(int(x) for x in s["lb"]) for s, _ in got} == expected, seed
        boxes, has0, _, st0 = run_streaming(tcn, fixpoint=fixpoint, debug=debug, stop_after_n_solutions=0, leaf_requires_assignment=0, **extra)
        assert st0["exhaustive"] and sum(box_volume(b) for b, _ in boxes) == len(expected), seed
        for b, _ in boxes:
            assert len(brute_force_solutions(b, props)) == box_volume(b), "every point of a solution box satisfies every propagator"


@pytest.mark.parametrize("chunk", range(2))
@pytest.mark.parametrize("fixpoint,debug", [(1, 0), (2, 0), (2, COMPACT)], ids=["wac1", "event", "event_compact"])
def test_random_models_tree_identical_under_the_gpu_rule(chunk, fixpoint, debug):
    from fuzz_models import random_model
    for seed in range(3000 + chunk * 25, 3000 + chunk * 25 + 25):
        tcn = frontend.Model.from_string(random_model(seed)).tcn()
        has_o, best_o, st_o = pyoracle.solve(tcn, subproblems_power=0, leaf_requires_assignment=1)
        has_g, best_g, st_g = capi.solve(tcn, capi.make_config(or_nodes=1, subproblems_power=0, timeout_ms=60000, fixpoint=fixpoint, debug=debug, leaf_requires_assignment=1))
        assert has_g == has_o and st_g["exhaustive"] == st_o["exhaustive"], seed
        for k in ("nodes", "fails", "solutions", "depth_max"):
            assert st_g[k] == st_o[k], (seed, k)
        if has_o:
            np.testing.assert_array_equal(best_g, best_o, err_msg=str(seed))
            if st_o["exhaustive"]:
                assert (best_g["lb"] == best_g["ub"]).all(), seed


@pytest.mark.parametrize("name,budget", [("accap_a3.fzn", 1_500_000), ("example_wordpress7_500.fzn", 1_500_000)])
def test_full_grid_paths_replay_under_the_gpu_rule(name, budget):
    """tests/test_gpu_fullgrid_paths.py with the gpu rule switched on: sampled workgroups of the production grid stand on stores the oracle reaches by replaying
    their paths under the same rule."""
    _, tcn, _ = preprocess.load_fzn_simplified(os.path.join(BENCH, name))
    s = capi.Session(tcn, capi.make_config(fixpoint=2, stop_after_n_nodes_total=budget, timeout_ms=300000, debug=KEEP, leaf_requires_assignment=1))
    plan = s.plan()
    s.start()
    while not s.poll()[1]:
        pass
    has, best, st = s.finish()
    assert st["nodes"] >= budget and not st["exhaustive"]
    if has:
        assert (best["lb"] == best["ub"]).all(), "a solution of the gpu rule is a full assignment"
    checked = compared = 0
    for wg in sorted(set(int(x) for x in np.linspace(0, plan["num_blocks"] - 1, 96))):
        hdr, dec = s.debug_path(wg)
        if not hdr["had_work"] or hdr["depth"] != hdr["decisions"]:
            continue
        last = s.debug_last_store(wg)
        store, failed, mismatch = pyoracle.replay_path(tcn, plan["subproblems_power"], hdr, dec, leaf_requires_assignment=1)
        assert mismatch == -1, f"{name} workgroup {wg}: the oracle does not take decision {mismatch} of {hdr}"
        assert failed == bool(hdr["last_node_failed"]), (name, wg, hdr)
        if not failed:
            np.testing.assert_array_equal(store, last, err_msg=f"{name} workgroup {wg} {hdr}")
            compared += 1
        checked += 1
    s.close()
    assert checked >= 40 and compared >= 12, (checked, compared)


LOOSE_FZN = ("var 1..3: x :: output_var;\nvar 1..3: y :: output_var;\nvar 0..2: z :: output_var;\nconstraint int_lt(x, y);\nsolve satisfy;\n")


def _cli(args, text, tmp_path):
    path = os.path.join(str(tmp_path), "m.fzn")
    with open(path, "w") as f:
        f.write(text)
    r = subprocess.run([TURBO, *args, path], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr
    return r.stdout


@pytest.mark.parametrize("simplify", [[], ["-disable_simplify"]], ids=["simplify", "disable_simplify"])
def test_cli_arch_selects_the_leaf_rule(tmp_path, simplify):
    """x < y over 1..3, z in 0..2 unconstrained.  Without the simplifier `-arch gpu -a` prints the 9 assignments and `-arch barebones -a` the 2 boxes its rule
    stops on.  The simplifier drops a variable nothing constrains (common_solving.hpp:537-585: useless-variable elimination; it is printed from its root
    domain), so the simplified network has the 3 (x, y) pairs as its full assignments."""
    out = _cli(["-arch", "gpu", "-a", "-s", *simplify], LOOSE_FZN, tmp_path)
    sols = re.findall(r"x = (\d);\ny = (\d);\nz = (\d);\n----------", out)
    pairs = sorted((str(x), str(y)) for x in (1, 2, 3) for y in (1, 2, 3) if x < y)
    if simplify:
        assert sorted(sols) == sorted((x, y, str(z)) for x, y in pairs for z in (0, 1, 2))
        assert "num_solutions=9" in out
    else:
        assert sorted((x, y) for x, y, _ in sols) == pairs and "num_solutions=3" in out
    assert "==========" in out
    out = _cli(["-arch", "barebones", "-a", "-s", *simplify], LOOSE_FZN, tmp_path)
    n = int(re.search(r"num_solutions=(\d+)", out).group(1))
    assert n == 2 and "==========" in out  # x = 1 with y in 2..3 open, x = 2 with y = 3 (and z open when it is in the store)
