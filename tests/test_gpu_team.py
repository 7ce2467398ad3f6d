"""Workgroup teams (store layout 5, kernels.hpp: solve_kernel_team; the plan for sweeps over stores too big for the hot tier, TB_TEAM=1 forces it on smaller ones): the workgroups of one XCD search ONE subproblem together on ONE store in global
memory -- partitioned sweeps, replicated control, the leader talks to the queue / grid words / host.  Same fixpoints, same trees:
  * TB_TEAM_ALL=1 makes the whole grid one team (whatever the XCDs: the protocol only uses agent-scope accesses), and one team walking 2^d subproblems in order must
    walk the ORACLE's tree, node for node, however many members share the sweeps (1, 3, 16, 40 workgroups);
  * full grids (eight teams racing, formed from the XCDs the workgroups really run on) prove the reference-held optima, enumerate exactly the solutions of satisfaction
    problems, and agree with the one-workgroup-per-subproblem kernels on the synthetic network.
The environment switches are read at session creation.
"""
import os

import numpy as np
import pytest

from conftest import BENCH, SLOW_FOR_ORACLE, known_answers
from oracle import pyoracle
from turbo_amd import capi, frontend

pytestmark = pytest.mark.gpu
ROWS = known_answers()
FAST = [r for r in ROWS if r[0] not in SLOW_FOR_ORACLE]
TEAM = dict(only_global_memory=1, threads_per_block=1024)  # what the team plan needs: the store in global memory, 1024-thread workgroups


@pytest.fixture
def team_env(monkeypatch):
    monkeypatch.setenv("TB_TEAM", "1")
    return monkeypatch


def plan_of(tcn, **cfg):
    s = capi.Session(tcn, capi.make_config(**cfg))
    p = s.plan()
    s.close()
    return p


@pytest.mark.parametrize("members", [1, 3, 16, 40])
@pytest.mark.parametrize("fixpoint", [1, 0], ids=["wac1", "ac1"])
@pytest.mark.parametrize("rel", ["test_data/sudoku_opt4.fzn", "test_data/pat2.fzn", "test_data/pat7.fzn", "test_data/pennies5.fzn", "accap_a3.fzn"])
def test_one_team_walks_the_oracles_tree(team_env, rel, fixpoint, members):
    team_env.setenv("TB_TEAM_ALL", "1")
    tcn = frontend.load_fzn(os.path.join(BENCH, rel))
    power, cut = 4, 3000
    has_o, best_o, st_o = pyoracle.solve(tcn, subproblems_power=power, cutnodes=cut)
    cfg = dict(or_nodes=members, subproblems_power=power, stop_after_n_nodes=cut, timeout_ms=120000, fixpoint=fixpoint, **TEAM)
    assert plan_of(tcn, **cfg)["kernel_opt"] == 10, "the team kernel was not planned"
    has_g, best_g, st_g = capi.solve(tcn, capi.make_config(**cfg))
    assert has_g == has_o
    for k in ("nodes", "fails", "solutions", "depth_max", "eps_solved_subproblems", "eps_skipped_subproblems"):
        assert st_g[k] == st_o[k], k
    if has_o:
        np.testing.assert_array_equal(best_g, best_o)


@pytest.mark.parametrize("members", [1, 5, 24])
@pytest.mark.parametrize("fixpoint", [1, 0], ids=["wac1", "ac1"])
def test_one_team_walks_the_oracles_tree_on_a_synthetic_network(team_env, fixpoint, members):
    """A small instance of the synthetic generator: mixed-class slices, products of non-negative operands.  That is the network on which the team kernel takes its short cuts
    -- the product rule with the narrowing pre-test before its divisions, slices handed out on demand (wac1), class-sorted 1024-record windows with the operands gathered a
    slice ahead (ac1) -- and the tree must still be the oracle's, node for node."""
    from turbo_amd.synth import make_synthetic
    team_env.setenv("TB_TEAM_ALL", "1")
    tcn = make_synthetic(3000, 14000, seed=11)
    power, cut = 3, 600
    has_o, best_o, st_o = pyoracle.solve(tcn, subproblems_power=power, cutnodes=cut)
    cfg = dict(or_nodes=members, subproblems_power=power, stop_after_n_nodes=cut, timeout_ms=120000, fixpoint=fixpoint, **TEAM)
    assert plan_of(tcn, **cfg)["kernel_opt"] == 10
    has_g, best_g, st_g = capi.solve(tcn, capi.make_config(**cfg))
    assert has_g == has_o
    for k in ("nodes", "fails", "solutions", "depth_max", "eps_solved_subproblems", "eps_skipped_subproblems"):
        assert st_g[k] == st_o[k], k
    if has_o:
        np.testing.assert_array_equal(best_g, best_o)


@pytest.mark.parametrize("members", [1, 3, 16, 40])
@pytest.mark.parametrize("rel", ["test_data/sudoku_opt4.fzn", "test_data/pat2.fzn", "test_data/pat7.fzn", "test_data/pennies5.fzn", "accap_a3.fzn"])
def test_one_team_shares_the_rounds_of_the_event_fixpoint_and_walks_the_oracles_tree(team_env, rel, members):
    """r06: the event-driven fixpoint of a team (kernels.hpp: fixpoint_event_team; TB_TEAM_EVENT=1, kernel_opt 5) -- dirty bitmaps in global memory, slices owned by the
    team's waves, one team barrier per round, the change list replicated.  One team of 1 / 3 / 16 / 40 workgroups (ownership tables for team sizes that are not powers of
    two) walks the ORACLE's tree node for node."""
    team_env.setenv("TB_TEAM_ALL", "1")
    team_env.setenv("TB_TEAM_EVENT", "1")
    tcn = frontend.load_fzn(os.path.join(BENCH, rel))
    power, cut = 4, 3000
    has_o, best_o, st_o = pyoracle.solve(tcn, subproblems_power=power, cutnodes=cut)
    cfg = dict(or_nodes=members, subproblems_power=power, stop_after_n_nodes=cut, timeout_ms=120000, fixpoint=2, **TEAM)
    plan = plan_of(tcn, **cfg)
    assert plan["kernel_opt"] == 5 and plan["kernel_event"] == 1, f"the event team kernel was not planned: {plan}"
    has_g, best_g, st_g = capi.solve(tcn, capi.make_config(**cfg))
    assert has_g == has_o
    for k in ("nodes", "fails", "solutions", "depth_max", "eps_solved_subproblems", "eps_skipped_subproblems"):
        assert st_g[k] == st_o[k], k
    if has_o:
        np.testing.assert_array_equal(best_g, best_o)


@pytest.mark.parametrize("members", [1, 5, 24])
def test_one_event_team_walks_the_oracles_tree_on_a_synthetic_network(team_env, members):
    from turbo_amd.synth import make_synthetic
    team_env.setenv("TB_TEAM_ALL", "1")
    team_env.setenv("TB_TEAM_EVENT", "1")
    tcn = make_synthetic(3000, 14000, seed=11)
    power, cut = 3, 600
    has_o, best_o, st_o = pyoracle.solve(tcn, subproblems_power=power, cutnodes=cut)
    cfg = dict(or_nodes=members, subproblems_power=power, stop_after_n_nodes=cut, timeout_ms=120000, fixpoint=2, **TEAM)
    assert plan_of(tcn, **cfg)["kernel_opt"] == 5
    has_g, best_g, st_g = capi.solve(tcn, capi.make_config(**cfg))
    assert has_g == has_o
    for k in ("nodes", "fails", "solutions", "depth_max", "eps_solved_subproblems", "eps_skipped_subproblems"):
        assert st_g[k] == st_o[k], k
    if has_o:
        np.testing.assert_array_equal(best_g, best_o)


@pytest.mark.parametrize("rel,expected", [r for r in FAST if r[0].split("/")[-1] in ("pat2.fzn", "pat7.fzn", "pennies5.fzn", "sudoku_opt4.fzn", "bug4.fzn", "pat11.fzn", "reified_in.fzn", "sudoku_opt_p0.fzn")])
def test_event_teams_of_the_real_xcds_prove_the_known_optima(team_env, rel, expected):
    team_env.setenv("TB_TEAM_EVENT", "1")
    tcn = frontend.load_fzn(os.path.join(BENCH, rel))
    has, best, st = capi.solve(tcn, capi.make_config(timeout_ms=120000, fixpoint=2, **TEAM))
    assert st["threads_per_block"] == 1024 and st["mem_kind"] == 0
    assert has and st["exhaustive"] == 1
    assert tcn.objective_of(best) == expected
    _, failed, ent, _, _ = pyoracle.propagate(best, tcn.props)
    assert not failed and ent
    assert st["eps_solved_subproblems"] + st["eps_skipped_subproblems"] == 1 << st["subproblems_power"], "every subproblem exactly once"


@pytest.mark.parametrize("window", [0, 128, 4096])
@pytest.mark.parametrize("fixpoint", [1, 0], ids=["wac1", "ac1"])
def test_record_windows_do_not_change_the_tree(team_env, fixpoint, window):
    """TB_GLOBAL_SORT_WINDOW: the records of a store in global memory class-sorted inside windows of W records (engine.hip: to_internal; 1024 by default for a team's plain
    sweeps, the caller's order otherwise).  The order of the records changes how long the fixpoint takes, never what it is: same tree as the oracle's for every W."""
    from turbo_amd.synth import make_synthetic
    team_env.setenv("TB_TEAM_ALL", "1")
    team_env.setenv("TB_GLOBAL_SORT_WINDOW", str(window))
    tcn = make_synthetic(3000, 14000, seed=12)
    power, cut = 2, 300
    has_o, best_o, st_o = pyoracle.solve(tcn, subproblems_power=power, cutnodes=cut)
    has_g, best_g, st_g = capi.solve(tcn, capi.make_config(or_nodes=7, subproblems_power=power, stop_after_n_nodes=cut, timeout_ms=120000, fixpoint=fixpoint, **TEAM))
    assert has_g == has_o
    for k in ("nodes", "fails", "solutions", "depth_max", "eps_solved_subproblems", "eps_skipped_subproblems"):
        assert st_g[k] == st_o[k], k
    if has_o:
        np.testing.assert_array_equal(best_g, best_o)


@pytest.mark.parametrize("rule", [0, 1], ids=["barebones_rule", "gpu_rule"])
def test_one_team_under_both_leaf_rules(team_env, rule):
    team_env.setenv("TB_TEAM_ALL", "1")
    from leaf_rule_models import as_tcn, loose_network
    for seed in range(8):
        rng = np.random.default_rng(4200 + seed)
        store, props = loose_network(rng)
        tcn = as_tcn(store, props, var_order=int(rng.integers(0, 5)), val_order=int(rng.integers(0, 4)))
        for power in (0, 3):
            has_o, best_o, st_o = pyoracle.solve(tcn, subproblems_power=power, stop_after_n_solutions=0, leaf_requires_assignment=rule)
            has_g, best_g, st_g = capi.solve(tcn, capi.make_config(or_nodes=5, subproblems_power=power, timeout_ms=60000, fixpoint=1, stop_after_n_solutions=0,
                                                                   leaf_requires_assignment=rule, **TEAM))
            assert has_g == has_o and st_g["exhaustive"] == st_o["exhaustive"], (seed, power)
            for k in ("nodes", "fails", "solutions", "depth_max", "eps_solved_subproblems", "eps_skipped_subproblems"):
                assert st_g[k] == st_o[k], (seed, power, k)


@pytest.mark.parametrize("fixpoint", [1, 0], ids=["wac1", "ac1"])
@pytest.mark.parametrize("rel,expected", FAST)
def test_teams_of_the_real_xcds_prove_the_known_optima(team_env, rel, expected, fixpoint):
    """The whole grid, teams formed from HW_REG_XCC_ID, racing through the queue with the incumbent exchanged through the grid words."""
    tcn = frontend.load_fzn(os.path.join(BENCH, rel))
    has, best, st = capi.solve(tcn, capi.make_config(timeout_ms=120000, fixpoint=fixpoint, **TEAM))
    assert st["threads_per_block"] == 1024 and st["mem_kind"] == 0
    assert has and st["exhaustive"] == 1
    assert tcn.objective_of(best) == expected
    _, failed, ent, _, _ = pyoracle.propagate(best, tcn.props)
    assert not failed and ent
    assert st["eps_solved_subproblems"] + st["eps_skipped_subproblems"] == 1 << st["subproblems_power"], "every subproblem exactly once"


@pytest.mark.parametrize("split,relaxed", [(1, 1), (2, 0), (2, 1), (4, 1)], ids=["relaxed", "two_per_xcd", "two_per_xcd_relaxed", "four_per_xcd_relaxed"])
@pytest.mark.parametrize("rel,expected", [r for r in FAST if r[0].split("/")[-1] in ("pat2.fzn", "pat7.fzn", "pennies5.fzn", "sudoku_opt4.fzn", "bug4.fzn", "pat11.fzn")])
def test_teams_per_xcd_and_barrier_flavours(team_env, rel, expected, split, relaxed):
    """TB_TEAM_SPLIT: two / four teams per XCD (the workgroups of an XCD dealt to them in turn); TB_TEAM_RELAXED: the barrier with relaxed agent-scope atomics and an explicit
    wait for the wave's own memory operations instead of acq_rel fences.  Same optima, every subproblem exactly once; and one of each walks the oracle's tree as ONE team."""
    team_env.setenv("TB_TEAM_SPLIT", str(split))
    team_env.setenv("TB_TEAM_RELAXED", str(relaxed))
    tcn = frontend.load_fzn(os.path.join(BENCH, rel))
    has, best, st = capi.solve(tcn, capi.make_config(timeout_ms=120000, fixpoint=1, **TEAM))
    assert has and st["exhaustive"] == 1 and tcn.objective_of(best) == expected
    assert st["eps_solved_subproblems"] + st["eps_skipped_subproblems"] == 1 << st["subproblems_power"]
    team_env.setenv("TB_TEAM_ALL", "1")
    has_o, best_o, st_o = pyoracle.solve(tcn, subproblems_power=3, cutnodes=2000)
    has_g, best_g, st_g = capi.solve(tcn, capi.make_config(or_nodes=24, subproblems_power=3, stop_after_n_nodes=2000, timeout_ms=120000, fixpoint=1, **TEAM))
    for k in ("nodes", "fails", "solutions", "depth_max", "eps_solved_subproblems", "eps_skipped_subproblems"):
        assert st_g[k] == st_o[k], k


@pytest.mark.parametrize("or_nodes", [1, 3, 8, 24])
def test_small_grids_of_real_xcd_teams_keep_to_their_own_slabs(team_env, or_nodes):
    """r05 indexed a team's store and snapshot stack by XCD x split + k -- up to 31 (63 with eight teams per XCD) -- while g_store / g_snap hold one slab per
    WORKGROUP: a grid smaller than that (`-or 8`) wrote past their end.  The slabs are now those of the team's leader.  Grids of 1, 3, 8 and 24 workgroups, teams formed from the
    XCDs they really land on (no TB_TEAM_ALL), eight teams per XCD too: the known optimum, every subproblem exactly once, and with one workgroup the oracle's tree."""
    rows = [r for r in FAST if r[0].split("/")[-1] in ("pat2.fzn", "pennies5.fzn", "sudoku_opt4.fzn")]
    for split in (4, 8):
        team_env.setenv("TB_TEAM_SPLIT", str(split))
        for rel, expected in rows:
            tcn = frontend.load_fzn(os.path.join(BENCH, rel))
            cfg = dict(or_nodes=or_nodes, subproblems_power=6, timeout_ms=120000, fixpoint=1, **TEAM)
            assert plan_of(tcn, **cfg)["kernel_opt"] == 10 and plan_of(tcn, **cfg)["num_blocks"] == or_nodes
            has, best, st = capi.solve(tcn, capi.make_config(**cfg))
            assert has and st["exhaustive"] == 1 and tcn.objective_of(best) == expected, (rel, split)
            assert st["eps_solved_subproblems"] + st["eps_skipped_subproblems"] == 64
            _, failed, ent, _, _ = pyoracle.propagate(best, tcn.props)
            assert not failed and ent
            if or_nodes == 1:
                has_o, best_o, st_o = pyoracle.solve(tcn, subproblems_power=6)
                for k in ("nodes", "fails", "solutions", "depth_max", "eps_solved_subproblems", "eps_skipped_subproblems"):
                    assert st[k] == st_o[k], (rel, k)
                np.testing.assert_array_equal(best, best_o)


_MID_SIZE_PROOF = []


def _oracle_proof_of_the_mid_size_network():
    if not _MID_SIZE_PROOF:  # (7 s of one host core, once for the three modes)
        tcn = frontend.load_fzn(os.path.join(BENCH, "example_wordpress7_500.fzn"))
        has, _, st = pyoracle.solve(tcn, subproblems_power=6, fixed_bound=260)
        _MID_SIZE_PROOF.append((has, st))
    return _MID_SIZE_PROOF[0]


@pytest.mark.parametrize("mode", ["wac1", "ac1", "event"])
def test_racing_teams_of_the_real_xcds_count_the_oracles_nodes_on_a_mid_size_network(team_env, mode):
    """VERDICT r05 item 3, last sentence: teams formed from the XCDs the workgroups really run on (no TB_TEAM_ALL; a full grid, four teams per XCD racing through the queue)
    against the oracle's COUNTERS on a mid-size network -- wordpress7_500 as parsed, 16 963 variables x 45 967 propagators, its store forced into global memory.  The search is
    the proof of `objective <= 260` over 2^6 subproblems (tb_config.use_fixed_bound: no incumbent is exchanged, so the tree does not depend on which team takes which
    subproblem when): 2 896 nodes, 1 288 of them failed, every subproblem refuted -- the same numbers from the sequential oracle, the sweeping teams and the event teams."""
    tcn = frontend.load_fzn(os.path.join(BENCH, "example_wordpress7_500.fzn"))
    if mode == "event":
        team_env.setenv("TB_TEAM_EVENT", "1")
    fixpoint = {"wac1": 1, "ac1": 0, "event": 2}[mode]
    cfg = dict(subproblems_power=6, use_fixed_bound=1, fixed_bound=260, timeout_ms=240000, fixpoint=fixpoint, **TEAM)
    plan = plan_of(tcn, **cfg)
    assert plan["kernel_opt"] == (5 if mode == "event" else 10) and plan["num_blocks"] >= 8, plan
    has_g, best_g, st_g = capi.solve(tcn, capi.make_config(**cfg))
    has_o, st_o = _oracle_proof_of_the_mid_size_network()
    assert not has_o and not has_g and st_g["exhaustive"] == 1
    assert st_o["nodes"] > 2000 and st_o["eps_solved_subproblems"] == 64
    for k in ("nodes", "fails", "solutions", "depth_max", "eps_solved_subproblems", "eps_skipped_subproblems"):
        assert st_g[k] == st_o[k], k


JOIN_SCRIPT = r"""
import os, sys, time
sys.path.insert(0, os.environ["TB_ROOT"])
from turbo_amd import capi, frontend
from turbo_amd.synth import make_synthetic
TEAM = dict(only_global_memory=1, threads_per_block=1024)
cus = capi.device_info(0)["compute_units"]
tcn = make_synthetic(30000, 120000, seed=7)
big = capi.Session(tcn, capi.make_config(fixpoint=1, timeout_ms=0, or_nodes=cus // 2, **TEAM))   # one 1024-thread workgroup on half of the CUs, until told to stop
os.environ["TB_TEAM_JOIN_MS"] = "1"
small = frontend.load_fzn(os.path.join(os.environ["TB_ROOT"], "benchmarks", "test_data", "pat7.fzn"))
late = capi.Session(small, capi.make_config(fixpoint=1, timeout_ms=0, **TEAM))                      # one workgroup per CU: half of them cannot become resident beside the first grid
assert late.plan()["kernel_opt"] == 10 and late.plan()["num_blocks"] == cus, late.plan()
print("STAGE sessions created", flush=True)
big.start()
time.sleep(0.5)
print("STAGE first search running", flush=True)
late.start()
print("STAGE second search launched", flush=True)
t0 = time.time()
while not late.poll()[1]:
    assert time.time() - t0 < 60, "the second team kernel hangs"
    time.sleep(0.01)
print("STAGE second kernel left after %.3f s" % (time.time() - t0), flush=True)
try:
    late.finish()
    print("RESULT no error", flush=True)
except capi.TurboHipError as e:
    print("RESULT", e, flush=True)
big.stop()   # (the second session is closed after the first search has ended: hipFree waits for every kernel of the device)
t0 = time.time()
while not big.poll()[1]:
    assert time.time() - t0 < 60, "the first search does not obey the stop request"
    time.sleep(0.01)
has, best, st = big.finish()
print("BIG nodes", st["nodes"], flush=True)
late.close()
big.close()
"""


def test_a_grid_that_never_becomes_resident_is_reported_not_waited_for():
    """team formation waits for every workgroup of the grid to register, and the launch is an ordinary one.  With the limit of the wait set to 1 ms
    (TB_TEAM_JOIN_MS) and a first search holding half of the CUs, the second session's team kernel (one workgroup per CU: half of them cannot become resident) must come
    back with TB_ERR_STATE "team formation failed" -- no hang, whatever timeout_ms is -- and the first search must be unharmed.  In a process of its own: a hang must not
    take the suite with it."""
    import subprocess
    import sys
    root = os.path.dirname(BENCH)
    try:
        p = subprocess.run([sys.executable, "-c", JOIN_SCRIPT], env=dict(os.environ, TB_ROOT=root, TB_TEAM="1"), capture_output=True, text=True, timeout=200)
    except subprocess.TimeoutExpired as e:
        pytest.fail("hung after: " + (e.stdout.decode() if isinstance(e.stdout, bytes) else str(e.stdout))[-600:])
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert "RESULT" in p.stdout and "team formation failed" in p.stdout, p.stdout[-2000:]
    big = [l for l in p.stdout.splitlines() if l.startswith("BIG nodes")]
    assert big and int(big[0].split()[-1]) > 0


def test_teams_enumerate_every_solution_once(team_env):
    from test_gpu_streaming import MANY, run_streaming
    tcn = frontend.Model.from_string(MANY).tcn()
    _, _, ost = pyoracle.solve(tcn, stop_after_n_solutions=0)
    got, has, best, st = run_streaming(tcn, fixpoint=1, stop_after_n_solutions=0, **TEAM)
    assert has and st["exhaustive"] and st["solutions"] == ost["solutions"] == len(got)
    assert len({s.tobytes() for s, _ in got}) == len(got)


def test_teams_are_the_default_plan_where_the_hot_tier_was(monkeypatch):
    """A store in global memory with more variables than the hot tier holds, sweeping fixpoints: the engine plans teams by itself (kernel_opt 10), the hot tier with TB_TEAM=0
    (kernel_opt 6), and the event fixpoint keeps its hot-tier kernel."""
    from turbo_amd.synth import make_synthetic
    monkeypatch.delenv("TB_TEAM", raising=False)
    tcn = make_synthetic(30000, 120000, seed=7)
    assert plan_of(tcn, fixpoint=1, timeout_ms=60000)["kernel_opt"] == 10
    assert plan_of(tcn, fixpoint=0, timeout_ms=60000)["kernel_opt"] == 10
    assert plan_of(tcn, fixpoint=2, threads_per_block=1024, timeout_ms=60000)["kernel_opt"] == 3
    monkeypatch.setenv("TB_TEAM", "0")
    assert plan_of(tcn, fixpoint=1, timeout_ms=60000)["kernel_opt"] == 6
    monkeypatch.delenv("TB_TEAM")
    # and the default plan searches: same optimum as the single-workgroup kernels on a budget-free small instance
    small = frontend.load_fzn(os.path.join(BENCH, "test_data", "pat7.fzn"))
    assert plan_of(small, fixpoint=1, timeout_ms=60000)["kernel_opt"] != 10, "small networks keep their LDS-resident kernels"


def test_synthetic_network_teams_agree_with_single_workgroups(team_env):
    """BASELINE.json configs[4] in small (20k x 100k): the same node budget per searcher gives the same tree statistics whether a searcher is a workgroup or a team."""
    from turbo_amd.synth import make_synthetic
    tcn = make_synthetic(20000, 100000, seed=42)
    cfg = dict(subproblems_power=6, stop_after_n_nodes=40, timeout_ms=240000, fixpoint=1, **TEAM)
    team_env.setenv("TB_TEAM_ALL", "1")
    has_t, best_t, st_t = capi.solve(tcn, capi.make_config(or_nodes=32, **cfg))
    team_env.setenv("TB_TEAM", "0")
    has_w, best_w, st_w = capi.solve(tcn, capi.make_config(or_nodes=1, **cfg))
    for k in ("nodes", "fails", "solutions", "depth_max", "eps_solved_subproblems", "eps_skipped_subproblems"):
        assert st_t[k] == st_w[k], k
    assert has_t == has_w
    if has_t:
        np.testing.assert_array_equal(best_t, best_w)
        _, failed, ent, _, _ = pyoracle.propagate(best_t, tcn.props)
        assert not failed and ent
