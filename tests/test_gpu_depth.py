"""Decision stacks that grow on demand (barebones_dive_and_solve.hpp:401-403 reallocates a block's vector when it is full).

The engine sizes the first segment of every workgroup's stack on the host (tb_config.decision_stack_depth, 16 384 by default);
a workgroup whose search goes deeper takes further segments -- up to 16 in all -- from a per-session pool, inside the kernel.
Beyond that the search ends with TB_ERR_DEPTH and tb_solve (and the CLI) run it again with 8x larger segments.
"""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from oracle import pyoracle
from turbo_amd import capi, frontend

pytestmark = pytest.mark.gpu
TURBO = os.path.join(ROOT, "turbo_amd", "bin", "turbo")


def chain(n):
    """x1 <= x2 <= ... <= xn over 0..1, input order / indomain_min: the first solution (all zero) sits n - 1 decisions deep,
    and with `minimize -sum`-like objective `xn` maximised the search comes back up and dives again."""
    decl = "".join(f"var 0..1: x{i};\n" for i in range(n))
    cons = "".join(f"constraint int_le(x{i},x{i + 1});\n" for i in range(n - 1))
    xs = ",".join(f"x{i}" for i in range(n))
    return decl + cons + f"solve :: int_search([{xs}], input_order, indomain_min, complete) maximize x{n - 1};\n"


def run_session(tcn, **kw):
    s = capi.Session(tcn, capi.make_config(or_nodes=1, subproblems_power=0, timeout_ms=60000, snapshot_levels=2, **kw))
    try:
        s.start()
        import time
        while not s.poll()[1]:
            time.sleep(0.001)
        return s.finish()
    finally:
        s.close()


@pytest.mark.parametrize("fixpoint", [1, 2], ids=["wac1", "event"])
def test_a_session_grows_its_decision_stack_in_the_kernel(fixpoint):
    tcn = frontend.Model.from_string(chain(150)).tcn()
    has_o, best_o, st_o = pyoracle.solve(tcn, subproblems_power=0)
    assert st_o["depth_max"] > 100
    # segments of 16 decisions: the search needs ten of them, through the session API (no host-side retry there)
    has, best, st = run_session(tcn, decision_stack_depth=16, fixpoint=fixpoint)
    assert has == has_o and st["exhaustive"] == 1
    assert (st["nodes"], st["fails"], st["solutions"], st["depth_max"]) == (st_o["nodes"], st_o["fails"], st_o["solutions"], st_o["depth_max"])
    np.testing.assert_array_equal(best, best_o)


def test_beyond_sixteen_segments_the_session_reports_depth_and_tb_solve_retries():
    tcn = frontend.Model.from_string(chain(400)).tcn()
    with pytest.raises(capi.TurboHipError) as e:
        run_session(tcn, decision_stack_depth=16)  # 16 x 16 = 256 decisions at most
    assert e.value.code == -6
    has_o, best_o, st_o = pyoracle.solve(tcn, subproblems_power=0)
    has, best, st = capi.solve(tcn, capi.make_config(or_nodes=1, subproblems_power=0, timeout_ms=60000, decision_stack_depth=16, snapshot_levels=2))
    assert has == has_o and st["exhaustive"] == 1 and st["depth_max"] == st_o["depth_max"]
    np.testing.assert_array_equal(best, best_o)


@pytest.mark.skipif(not os.path.exists(TURBO), reason="turbo CLI not built")
def test_the_cli_session_path_survives_a_deep_search(tmp_path):
    """`-i` goes through the session API (streaming): 20 000 decisions deep with the default 16 384-decision segments."""
    f = tmp_path / "deep.fzn"
    f.write_text(chain(20000))
    r = subprocess.run([TURBO, "-s", "-i", "-t", "120000", "-or", "2", "-sub", "1", str(f)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "==========" in r.stdout and "objective=1" in r.stdout
