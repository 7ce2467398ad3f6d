"""Store-level parity of the kernels of GLOBAL stores at BASELINE size: BASELINE.json configs[4], the synthetic
100 000 x 500 000 network, searched with the ENGINE'S OWN plan --

  * `-fp wac1` / `-fp ac1`: workgroup teams (kernel_opt 10: four teams per XCD formed from HW_REG_XCC_ID, relaxed team barrier, slices handed out on
    demand, for AC1 the 1024-record class-sort windows with the operands gathered a slice ahead), 32 teams racing through the queue;
  * `-fp event`: the hot tier (kernel_opt 3), one 1024-thread workgroup per subproblem --

under a node budget, then the ORACLE replays the path each sampled searcher stood on (oracle.c: orc_replay_path -- root, dive along the bits of the
subproblem index with its own variable selection, the recorded decisions each checked against the decision the oracle takes, the last node) and the
store under the last node must be the oracle's, bit for bit.  The same argument as tests/test_gpu_fullgrid_paths.py: which searcher gets which
subproblem is timing, the store under a given node is not.  For a team the path is its leader's (tb_session_debug_path reports "no work" for the other
members, who hold copies of the same decision stack) and the store is the team's shared store as the leader copied it out (g_last).
The replays (40 oracle nodes of 500 000 propagators each, ~4 s a path) run in a thread pool: ctypes releases the GIL and orc_replay_path is re-entrant.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from oracle import pyoracle
from turbo_amd import capi

pytestmark = pytest.mark.gpu
KEEP = 0x800000  # tb_config.reserved[0]: keep every workgroup's last store and path


@pytest.fixture(scope="module")
def net():
    from turbo_amd.synth import make_synthetic
    return make_synthetic(100_000, 500_000, seed=42)


def replay_sampled(net, s, plan, searchers, want: int):
    """Replay up to `want` of the searchers (workgroup indices) that were still searching -- those whose last node did not fail first, since only under such a node is the
    store defined and compared; returns (checked, compared stores, deepest path)."""
    live = []
    for wg in searchers:
        hdr, dec = s.debug_path(wg)
        if not hdr["had_work"] or hdr["depth"] != hdr["decisions"] or hdr["nodes"] == 0:
            continue
        live.append((wg, hdr, dec))
    live.sort(key=lambda c: bool(c[1]["last_node_failed"]))  # (stable: workgroup order inside each half)
    cand = [(wg, hdr, dec, s.debug_last_store(wg)) for wg, hdr, dec in live[:want]]
    with ThreadPoolExecutor(max_workers=min(12, os.cpu_count() or 1)) as pool:
        outs = list(pool.map(lambda c: pyoracle.replay_path(net, plan["subproblems_power"], c[1], c[2]), cand))
    compared = deepest = 0
    for (wg, hdr, dec, last), (store, failed, mismatch) in zip(cand, outs):
        assert mismatch == -1, f"workgroup {wg}: the oracle does not take decision {mismatch} of {hdr}"
        assert failed == bool(hdr["last_node_failed"]), f"workgroup {wg}: failed flag of the last node ({hdr})"
        if not failed:
            np.testing.assert_array_equal(store, last, err_msg=f"workgroup {wg} {hdr}")
            compared += 1
        deepest = max(deepest, hdr["depth"])
    return len(cand), compared, deepest


@pytest.mark.parametrize("fixpoint", [1, 0], ids=["wac1", "ac1"])
def test_teams_of_the_real_xcds_stand_on_the_oracles_stores_at_full_size(net, fixpoint, monkeypatch):
    for k in ("TB_TEAM", "TB_TEAM_ALL", "TB_TEAM_SPLIT", "TB_TEAM_RELAXED", "TB_GLOBAL_SORT_WINDOW"):
        monkeypatch.delenv(k, raising=False)  # the engine's own plan
    s = capi.Session(net, capi.make_config(fixpoint=fixpoint, stop_after_n_nodes_total=2600, timeout_ms=300000, debug=KEEP))
    plan = s.plan()
    info = capi.device_info(0)
    if info["compute_units"] == 256:  # an MI355X: the plan BENCH prints for configs[4]
        assert plan["kernel_opt"] == 10 and plan["num_blocks"] == 256 and plan["threads_per_block"] == 1024 and plan["mem_kind"] == 0, plan
    assert plan["kernel_opt"] == 10, "the team kernel was not planned"
    s.start()
    while not s.poll()[1]:
        pass
    has, best, st = s.finish()
    assert st["nodes"] >= 2600 and not st["exhaustive"], "the budget must end the search"
    leaders = [wg for wg in range(plan["num_blocks"]) if s.debug_path(wg)[0]["nodes"] > 0]
    assert 8 <= len(leaders) <= 64, f"{len(leaders)} teams"  # four per XCD on an MI355X: 32
    checked, compared, deepest = replay_sampled(net, s, plan, leaders, 32)
    s.close()
    # (every team of the grid is replayed: decisions and failed flags of all of them; the last node of about half of them has failed -- the store of a failed node is not
    #  defined, and how many teams stand on one when the budget ends is timing: the floor on the compared stores is kept low so that an unlucky stop does not fail the suite)
    assert checked >= 16 and compared >= 4, f"{checked} teams replayed, {compared} stores compared"
    print(f"synthetic 100k x 500k, {'wac1' if fixpoint else 'ac1'} in teams: {len(leaders)} teams, {checked} replayed, {compared} stores identical to the oracle's, deepest path {deepest}")


def test_hot_tier_event_kernel_stands_on_the_oracles_stores_at_full_size(net, monkeypatch):
    monkeypatch.delenv("TB_NO_HOT_TIER", raising=False)
    s = capi.Session(net, capi.make_config(fixpoint=2, stop_after_n_nodes_total=9000, timeout_ms=300000, debug=KEEP))
    plan = s.plan()
    info = capi.device_info(0)
    if info["compute_units"] == 256:
        assert plan["kernel_opt"] == 3 and plan["kernel_event"] == 1 and plan["num_blocks"] == 256 and plan["threads_per_block"] == 1024 and plan["mem_kind"] == 0, plan
    s.start()
    while not s.poll()[1]:
        pass
    has, best, st = s.finish()
    assert st["nodes"] >= 9000 and not st["exhaustive"]
    checked, compared, deepest = replay_sampled(net, s, plan, range(plan["num_blocks"]), 24)  # (of the 256 workgroups, the 24 first whose last node stands)
    s.close()
    assert checked >= 16 and compared >= 16, f"{checked} workgroups replayed, {compared} stores compared"
    print(f"synthetic 100k x 500k, event on the hot tier: {checked} workgroups replayed, {compared} stores identical to the oracle's, deepest path {deepest}")


def test_event_teams_of_the_real_xcds_stand_on_the_oracles_stores_at_full_size(net, monkeypatch):
    """The event fixpoint shared by the workgroups of a team (r06, TB_TEAM_EVENT=1: kernel_opt 5 -- dirty bitmaps in global memory, slices owned by the team's 128 waves, one
    team barrier per round): the 32 real-XCD teams of the full-size network under a node budget, every team's path replayed by the oracle."""
    for k in ("TB_TEAM", "TB_TEAM_ALL", "TB_TEAM_SPLIT", "TB_TEAM_RELAXED", "TB_GLOBAL_SORT_WINDOW"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("TB_TEAM_EVENT", "1")
    s = capi.Session(net, capi.make_config(fixpoint=2, threads_per_block=1024, stop_after_n_nodes_total=2600, timeout_ms=300000, debug=KEEP))
    plan = s.plan()
    assert plan["kernel_opt"] == 5 and plan["kernel_event"] == 1 and plan["threads_per_block"] == 1024 and plan["mem_kind"] == 0, plan
    s.start()
    while not s.poll()[1]:
        pass
    has, best, st = s.finish()
    assert st["nodes"] >= 2600 and not st["exhaustive"], "the budget must end the search"
    leaders = [wg for wg in range(plan["num_blocks"]) if s.debug_path(wg)[0]["nodes"] > 0]
    assert 8 <= len(leaders) <= 64, f"{len(leaders)} teams"
    checked, compared, deepest = replay_sampled(net, s, plan, leaders, 32)
    s.close()
    # (all 32 teams are replayed: decisions, failed flags -- most of them stand on a failed node when the budget ends -- and the stores of those that do not)
    assert checked >= 16 and compared >= 4, f"{checked} teams replayed, {compared} stores compared"
    print(f"synthetic 100k x 500k, event fixpoint in teams: {len(leaders)} teams, {checked} replayed, {compared} stores identical to the oracle's, deepest path {deepest}")
