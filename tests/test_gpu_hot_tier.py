"""Hot tier of a store in global memory (kernels.hpp: layout 3; engine.hip: plan_launch / renumber_by_reads): the 1024-thread search kernels of a network
too large for LDS keep the HOT_VARS most-read intervals in LDS and the rest in the workgroup's slab in global memory.  Same search, bit for bit:
one workgroup walks the oracle's tree and stops on its store; sampled workgroups of a full grid stand on stores the oracle reproduces by replaying their
paths; and the plan without the tier (TB_NO_HOT_TIER) gives the same counters."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from oracle import pyoracle
from turbo_amd import capi
from turbo_amd.synth import make_synthetic

pytestmark = pytest.mark.gpu
KEEP = 0x800000


@pytest.fixture(autouse=True)
def no_teams(monkeypatch):
    # r05: the sweeps of a network this size are planned in workgroup teams (tests/test_gpu_team.py); TB_TEAM=0 gives the hot tier this file is about
    monkeypatch.setenv("TB_TEAM", "0")


@pytest.fixture(scope="module")
def net():
    return make_synthetic(30_000, 120_000, seed=7)


def run(tcn, **kw):
    s = capi.Session(tcn, capi.make_config(timeout_ms=240000, debug=KEEP, **kw))
    plan = s.plan()
    s.start()
    while not s.poll()[1]:
        pass
    has, best, st = s.finish()
    return s, plan, has, best, st


@pytest.mark.parametrize("fixpoint,threads,opt", [(1, 0, 6), (0, 0, 6), (2, 1024, 3)], ids=["wac1", "ac1", "event"])
@pytest.mark.parametrize("power", [0, 3])
def test_one_workgroup_walks_the_oracles_tree(net, fixpoint, threads, opt, power):
    cut = 120
    has_o, best_o, st_o, trace, last_o = pyoracle.solve_traced(net, cut, power)
    s, plan, has_g, best_g, st_g = run(net, fixpoint=fixpoint, threads_per_block=threads, or_nodes=1, subproblems_power=power, stop_after_n_nodes=cut)
    assert plan["mem_kind"] == 0 and plan["threads_per_block"] == 1024 and plan["kernel_opt"] == opt, plan
    last_g = s.debug_last_store(0)
    s.close()
    assert has_g == has_o
    for k in ("nodes", "fails", "solutions", "depth_max", "eps_solved_subproblems", "eps_skipped_subproblems"):
        assert st_g[k] == st_o[k], k
    if has_o:
        np.testing.assert_array_equal(best_g, best_o)
    if not trace[-1]:
        np.testing.assert_array_equal(last_g, last_o)


def test_full_grid_paths_replay_on_the_oracle(net):
    s, plan, has, best, st = run(net, fixpoint=1, stop_after_n_nodes_total=6000)
    assert plan["kernel_opt"] == 6 and plan["mem_kind"] == 0
    checked = 0
    for wg in sorted(set(int(x) for x in np.linspace(0, plan["num_blocks"] - 1, 24))):
        hdr, dec = s.debug_path(wg)
        if not hdr["had_work"] or hdr["depth"] != hdr["decisions"]:
            continue
        last = s.debug_last_store(wg)
        store, failed, mismatch = pyoracle.replay_path(net, plan["subproblems_power"], hdr, dec)
        assert mismatch == -1, (wg, hdr, mismatch)
        assert failed == bool(hdr["last_node_failed"]), (wg, hdr)
        if not failed:
            np.testing.assert_array_equal(store, last, err_msg=str((wg, hdr)))
        checked += 1
    s.close()
    assert checked >= 12


def test_same_counters_without_the_tier(net):
    """TB_NO_HOT_TIER (read when the session is planned): plain store in global memory, same tree."""
    code = ("import json, os, sys; sys.path.insert(0, os.environ['TB_ROOT']);"
            "from turbo_amd import capi; from turbo_amd.synth import make_synthetic;"
            "t = make_synthetic(30_000, 120_000, seed=7);"
            "s = capi.Session(t, capi.make_config(timeout_ms=240000, fixpoint=1, or_nodes=1, subproblems_power=3, stop_after_n_nodes=120)); p = s.plan(); s.start();\n"
            "while not s.poll()[1]: pass\n"
            "st = s.finish()[2]; print(json.dumps([p['kernel_opt']] + [st[k] for k in ('nodes', 'fails', 'solutions', 'depth_max', 'num_deductions')]))")
    rows = []
    for env in ({}, {"TB_NO_HOT_TIER": "1"}):
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TB_ROOT=ROOT, **env), capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        rows.append(json.loads(p.stdout.strip().splitlines()[-1]))
    assert rows[0][0] == 6 and rows[1][0] == 0
    assert rows[0][1:5] == rows[1][1:5]
