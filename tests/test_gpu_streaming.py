"""Solution streaming (`-i`, `-a`, satisfaction `-n k`): the `gpu` path's producer/consumer protocol
(gpu_dive_and_solve.hpp:100-132,334-345) through tb_session_next_solution and through the CLI.

Checked against the oracle: the number of solution leaves of a satisfaction problem (the search tree is the
same tree, cut into EPS subproblems), validity of every streamed store on the network, and -- for optimisation --
that the improving sequence ends on the known optimum.
"""
import os
import re
import subprocess
import time

import numpy as np
import pytest

from conftest import BENCH, ROOT
from oracle import pyoracle
from turbo_amd import capi, frontend

pytestmark = pytest.mark.gpu
TURBO = os.path.join(ROOT, "turbo_amd", "bin", "turbo")

ALLDIFF3 = ("var 1..3: x :: output_var;\nvar 1..3: y :: output_var;\nvar 1..3: z :: output_var;\n"
            "constraint int_ne(x, y);\nconstraint int_ne(y, z);\nconstraint int_ne(x, z);\nsolve satisfy;\n")
# 5 variables over 0..3 with a sum and an ordering: a few hundred solutions, more than the ring holds
MANY = ("array [1..5] of var 0..3: q :: output_array([1..5]);\n"
        "constraint int_lin_le([1,1,1,1,1], [q[1],q[2],q[3],q[4],q[5]], 9);\n"
        "constraint int_le(q[1], q[2]);\nconstraint int_ne(q[3], q[4]);\nsolve satisfy;\n")


def run_streaming(tcn, timeout_s=60, **cfg):
    s = capi.Session(tcn, capi.make_config(stream_solutions=1, **cfg))
    s.start()
    got = []
    t0 = time.time()
    done = False
    while not done:
        _, done = s.poll()
        while (nxt := s.next_solution()) is not None:
            got.append(nxt)
        if time.time() - t0 > timeout_s:
            s.stop()
        time.sleep(0.0005)
    while (nxt := s.next_solution()) is not None:
        got.append(nxt)
    has, best, st = s.finish()
    s.close()
    return got, has, best, st


def assert_valid(tcn, store):
    out, failed, entailed, _, _ = pyoracle.propagate(store, tcn.props)
    assert not failed and entailed
    np.testing.assert_array_equal(out, store)


@pytest.mark.parametrize("text", [ALLDIFF3, MANY], ids=["alldiff3", "many"])
@pytest.mark.parametrize("fixpoint,debug", [(1, 0), (2, 0), (2, 0x100000)], ids=["wac1", "event", "event_compact"])
def test_all_solutions_of_a_satisfaction_problem(text, fixpoint, debug):
    tcn = frontend.Model.from_string(text).tcn()
    _, _, ost = pyoracle.solve(tcn, stop_after_n_solutions=0)
    assert ost["exhaustive"] and ost["solutions"] > 0
    got, has, best, st = run_streaming(tcn, fixpoint=fixpoint, stop_after_n_solutions=0, debug=debug)
    assert has and st["exhaustive"]
    assert st["solutions"] == ost["solutions"] == len(got)
    seen = set()
    for store, _ in got:
        assert_valid(tcn, store)
        seen.add(store.tobytes())
    assert len(seen) == len(got), "a solution leaf was handed over twice"
    if text is ALLDIFF3:
        assert len(got) == 6


@pytest.mark.parametrize("k", [1, 3, 7])
def test_first_k_solutions(k):
    tcn = frontend.Model.from_string(MANY).tcn()
    got, has, best, st = run_streaming(tcn, stop_after_n_solutions=k)
    assert has and not st["exhaustive"]
    assert len(got) == k  # racing workgroups may find more; only the first k tickets are handed over
    for store, _ in got:
        assert_valid(tcn, store)


@pytest.mark.parametrize("rel,expected", [("test_data/pennies5.fzn", 5), ("test_data/pat9.fzn", 19), ("test_data/sudoku_opt_p0.fzn", -3)])
@pytest.mark.parametrize("fixpoint,debug", [(1, 0), (2, 0), (2, 0x100000)], ids=["wac1", "event", "event_compact"])
def test_improving_solutions_end_on_the_optimum(rel, expected, fixpoint, debug):
    tcn = frontend.load_fzn(os.path.join(BENCH, rel))
    got, has, best, st = run_streaming(tcn, fixpoint=fixpoint, debug=debug)
    assert has and st["exhaustive"] and tcn.objective_of(best) == expected
    assert got, "an optimisation run that found a solution streams at least one"
    objs = [o for _, o in got]
    assert min(objs) == int(best[tcn.obj_var]["lb"])  # the optimum itself was handed over
    for store, o in got:
        assert int(store[tcn.obj_var]["lb"]) == o
        assert_valid(tcn, store)
    # every handed-over solution improved the incumbent of the device when it was found: no value twice
    assert len(set(objs)) == len(objs)


def test_session_without_streaming_never_hands_anything_over():
    tcn = frontend.load_fzn(os.path.join(BENCH, "test_data/sudoku_opt2.fzn"))
    s = capi.Session(tcn, capi.make_config())
    s.start()
    done = False
    while not done:
        _, done = s.poll()
        assert s.next_solution() is None
    has, _, st = s.finish()
    assert has and st["exhaustive"]
    s.close()


def _cli(args, text=None, path=None, tmp_path=None):
    if text is not None:
        path = os.path.join(tmp_path, "m.fzn")
        with open(path, "w") as f:
            f.write(text)
    r = subprocess.run([TURBO, *args, path], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr
    return r.stdout


def test_cli_all_solutions(tmp_path):
    out = _cli(["-a", "-s"], text=ALLDIFF3, tmp_path=str(tmp_path))
    sols = set(re.findall(r"x = (\d);\ny = (\d);\nz = (\d);\n----------", out))
    assert sols == {("1", "2", "3"), ("1", "3", "2"), ("2", "1", "3"), ("2", "3", "1"), ("3", "1", "2"), ("3", "2", "1")}
    assert out.count("----------") == 6 and "==========" in out
    assert "num_solutions=6" in out
    assert "WARNING" not in out


def test_cli_n_solutions(tmp_path):
    out = _cli(["-n", "4"], text=MANY, tmp_path=str(tmp_path))
    assert out.count("----------") == 4
    assert "==========" not in out  # the search was cut: not exhaustive
    out = _cli([], text=MANY, tmp_path=str(tmp_path))  # default -n 1: only the final solution, as barebones
    assert out.count("----------") == 1


def test_cli_intermediate_solutions_of_an_optimisation_problem():
    out = _cli(["-i", "-s", "-t", "60000"], path=os.path.join(BENCH, "test_data", "pat9.fzn"))
    assert "objective=19" in out and "==========" in out
    n = out.count("----------")
    assert n >= 1
    # the last solution printed is the optimal one: printed once, not repeated at the end
    blocks = out.split("----------")
    assert len(blocks) == n + 1
