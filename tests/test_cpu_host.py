"""CPU tests (no GPU): the C-ABI library loads and exports every declared symbol, host arithmetic,
the multi-GPU protocol over gloo (world_size 2), the front-end, and the CLI's error behaviour."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import BENCH, ROOT
from turbo_amd import capi, frontend

TURBO = os.path.join(ROOT, "turbo_amd", "bin", "turbo")


def declared_symbols(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(t[bf]_[a-z_0-9]+)\s*\(", text)))


def test_hip_library_exports_every_declared_symbol():
    L = capi.lib()
    syms = declared_symbols("turbo_hip.h")
    assert set(capi.EXPORTS) <= set(syms)
    for name in syms:
        assert hasattr(L, name), f"libturbo_hip.so does not export {name}"
    assert b"gfx950" in L.tb_version()


def test_front_library_exports_every_declared_symbol():
    L = frontend.lib()
    for name in declared_symbols("turbo_front.h"):
        assert hasattr(L, name), f"libturbo_front.so does not export {name}"


def test_struct_layouts_match_header():
    # sizes computed from include/turbo_hip.h by hand: tb_config = 7*8 + 21*4 = 140 -> 144 (8-byte alignment),
    # tb_stats = 9*8 + 11*8 + 3*8 + 12*4 + 6*8 + 4*8 = 312.  (r05: leaf_requires_assignment took tb_config's 4 bytes of tail padding, prof_ns[4] were appended to tb_stats)
    assert ctypes.sizeof(capi.TbConfig) == 144
    assert capi.TbConfig.leaf_requires_assignment.offset == 140
    assert ctypes.sizeof(capi.TbStats) == 312


def test_no_device_is_a_loud_error():
    if capi.lib().tb_device_count() > 0:
        pytest.skip("a GPU is present")
    tcn = frontend.load_fzn(os.path.join(BENCH, "test_data", "sudoku_opt2.fzn"))
    with pytest.raises(capi.TurboHipError) as e:
        capi.solve(tcn, capi.make_config(timeout_ms=1000))
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)
    with pytest.raises(capi.TurboHipError):
        capi.propagate(tcn.props, tcn.store[None, :])


@pytest.mark.parametrize("d,k,world", [(0, 0, 1), (3, 0, 2), (10, 0, 8), (10, 3, 8), (11, 2, 3), (13, 5, 7), (6, 9, 4), (17, 0, 3)])
def test_block_cyclic_shares_partition_the_index_space(d, k, world):
    """Every subproblem index belongs to exactly one rank; a rank's local numbering is in increasing global order;
    chunk c of 2^k consecutive indices belongs to rank c % world (include/turbo_hip.h: tb_eps_global_index)."""
    kk = min(k, d)
    seen = np.full(2 ** d, -1, dtype=np.int64)
    counts = []
    for r in range(world):
        n = capi.eps_local_count(d, k, r, world)
        counts.append(n)
        idx = [capi.eps_global_index(d, k, r, world, j) for j in range(n)]
        assert idx == sorted(idx)
        for g in idx:
            assert seen[g] == -1 and (g >> kk) % world == r
            seen[g] = r
        with pytest.raises(capi.TurboHipError):
            capi.eps_global_index(d, k, r, world, n)
    assert (seen >= 0).all() and sum(counts) == 2 ** d
    assert max(counts) - min(counts) <= 2 ** kk


def test_frontend_lowering_conventions():
    tcn = frontend.Model.from_string(
        "var 0..10: x; var 0..10: y; var bool: b; constraint int_lin_le([2,-3],[x,y],4); "
        "constraint int_le_reif(x,y,b); solve maximize x;").tcn()
    # constants 0,1,2 are pre-interned as variables 0,1,2 (common_solving.hpp:521)
    for v in range(3):
        assert tcn.store[v]["lb"] == tcn.store[v]["ub"] == v
    ops = {frontend.OP_NAMES[o] for o in tcn.props["op"]}
    assert ops <= {"ADD", "MUL", "LEQ", "EQ"}
    # maximize is rewritten to minimising a negated variable (common_solving.hpp:489-510)
    assert tcn.goal == 2 and tcn.obj_var != tcn.goal_var and tcn.obj_var >= 0
    assert tcn.strat_var_order[-1] == 1 and tcn.strat_off[-1] == tcn.strat_off[-2]  # default first_fail over the whole store


def test_frontend_errors():
    with pytest.raises(ValueError):
        frontend.Model.from_string("var 1..3: x; constraint nope(x); solve satisfy;")
    with pytest.raises(ValueError):
        frontend.Model.from_string("var 1..3: x constraint")
    with pytest.raises(ValueError):
        frontend.load_fzn(os.path.join(BENCH, "does_not_exist.fzn"))


def test_headline_instances_lower():
    shapes = {"example_wordpress7_500.fzn": (16963, 45967), "accap_a3.fzn": (933, 1060), "trains15.fzn": (24269, 24014)}
    for name, (v, p) in shapes.items():
        tcn = frontend.load_fzn(os.path.join(BENCH, name))
        assert (tcn.n_vars, tcn.n_props) == (v, p)
        assert int(tcn.props["x"].max()) < v and int(tcn.props["op"].max()) < 8


# ---- multi-process protocol over gloo ---------------------------------------------------------------

_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["TB_ROOT"])
import torch.distributed as dist
from turbo_amd.distributed import exchange_until_done, reduce_results, PINF

class FakeSession:
    """Scripted device: rank r finds bound b at poll number k, finishes at poll number `end`."""
    def __init__(self, script, end):
        self.script, self.end, self.n, self.best, self.pushed = script, end, 0, PINF, []
    def poll(self):
        self.n += 1
        for k, b in self.script:
            if self.n >= k: self.best = min(self.best, b)
        return self.best, self.n >= self.end
    def push_bound(self, b): self.pushed.append(b)
    def stop(self): pass

dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
r = dist.get_rank()
s = FakeSession([(2, 50), (5, 30)] if r == 0 else [(3, 40), (9, 12)], end=6 if r == 0 else 11)
gbest, rounds = exchange_until_done(s, dist, period_s=0.0)
assert gbest == 12, gbest
assert s.pushed == sorted(s.pushed, reverse=True) and s.pushed[-1] == 12, s.pushed   # monotone import of the incumbent
assert rounds >= 11                                                                  # nobody leaves before the slowest rank is done
winner, gb, tot = reduce_results(True, 30 if r == 0 else 12, {"nodes": 10 * (r + 1), "num_deductions": 7}, dist)
assert (winner, gb, tot["nodes"], tot["num_deductions"]) == (1, 12, 30, 14), (winner, gb, tot)
winner, gb, _ = reduce_results(True, 5, {}, dist)      # tie on the bound, no subproblem reported -> lowest rank
assert (winner, gb) == (0, 5)
# tie on the bound: the LOWEST SUBPROBLEM INDEX wins, whichever rank holds it (block-cyclic shares + work stealing: rank 1
# may well hold index 2 while rank 0 holds index 7) -- the canonical pass is only deterministic that way
winner, gb, _ = reduce_results(True, 5, {"best_subproblem": 7 if r == 0 else 2}, dist)
assert (winner, gb) == (1, 5), (winner, gb)
winner, gb, _ = reduce_results(True, 5 if r == 0 else 4, {"best_subproblem": 0 if r == 0 else 9}, dist)   # the bound comes first
assert (winner, gb) == (1, 4), (winner, gb)
winner, gb, _ = reduce_results(r == 0, 5, {"best_subproblem": 7}, dist)   # only rank 0 has a solution
assert (winner, gb) == (0, 5), (winner, gb)
winner, gb, _ = reduce_results(False, 0, {}, dist)
assert winner == -1
dist.destroy_process_group()
print("ok", r)
'''


def test_bound_exchange_protocol_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, TB_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "ok" in o


# ---- CLI error behaviour (src/config.cpp:128-220, src/turbo.cpp:22-56) --------------------------------

def run_turbo(*args):
    return subprocess.run([TURBO, *args], capture_output=True, text=True, timeout=60)


@pytest.mark.skipif(not os.path.exists(TURBO), reason="turbo CLI not built")
class TestCli:
    def test_unknown_arch(self):
        r = run_turbo("-arch", "tpu", "x.fzn")
        assert r.returncode == 1 and "Unknown architecture -arch tpu" in r.stderr

    def test_unknown_fixpoint(self):
        r = run_turbo("-fp", "ac3", "x.fzn")
        assert r.returncode == 1 and "Unknown fixpoint -fp ac3" in r.stderr

    def test_or_and_p_are_exclusive(self):
        r = run_turbo("-or", "4", "-p", "4", "x.fzn")
        assert r.returncode == 1 and "cannot be used at the same time" in r.stderr

    def test_eps_orders_go_together(self):
        r = run_turbo("-eps_var_order", "input_order", os.path.join(BENCH, "test_data", "sudoku_opt2.fzn"))
        assert r.returncode == 1 and "must be specified together" in r.stdout

    def test_unparsable_input(self, tmp_path):
        f = tmp_path / "bad.fzn"
        f.write_text("var 1..3 x;")
        r = run_turbo(str(f))
        assert r.returncode == 1 and "Could not parse input file." in r.stderr

    def test_cpu_arch_is_refused_not_faked(self):
        r = run_turbo("-arch", "cpu", os.path.join(BENCH, "test_data", "sudoku_opt2.fzn"))
        assert r.returncode == 1 and "not provided by this build" in r.stderr

    def test_unsat_at_interpretation(self):
        r = run_turbo("-s", os.path.join(BENCH, "unsolved_bugs_data", "false.fzn"))
        assert r.returncode == 0 and "=====UNSATISFIABLE=====" in r.stdout
        assert '%%%mzn-stat: command_line="' in r.stdout and "%%%mzn-stat-end" in r.stdout

    def test_command_line_echo_defaults(self):
        r = run_turbo("-s", os.path.join(BENCH, "unsolved_bugs_data", "false.fzn"))
        line = r.stdout.splitlines()[0]
        for part in ("-t 0", "-n 1", "-arch barebones", "-or 0", "-sub -1", "-subfactor 300", "-fp auto",
                     "-seed 0", "-eps_var_order default", "-cutnodes 0"):
            assert part in line, part


# ---- synthetic workload (BASELINE.json configs[4]) ----------------------------------------------------

def test_synthetic_network_is_satisfiable_by_construction():
    from oracle import pyoracle
    from turbo_amd.synth import make_synthetic
    tcn = make_synthetic(2000, 10000, seed=42)
    assert tcn.n_vars == 2003 and tcn.n_props == 10000
    v = tcn.hidden_solution
    p = tcn.props
    x, y, z = v[p["x"]], v[p["y"]], v[p["z"]]
    ok = np.select([p["op"] == 0, p["op"] == 1, p["op"] == 4, p["op"] == 5, p["op"] == 6, p["op"] == 7],
                   [x == y + z, x == y * z, x == np.minimum(y, z), x == np.maximum(y, z), x == (y == z), x == (y <= z)])
    assert ok.all()
    assert ((tcn.store["lb"] <= v) & (v <= tcn.store["ub"])).all()
    # deterministic
    again = make_synthetic(2000, 10000, seed=42)
    assert np.array_equal(again.props, tcn.props) and np.array_equal(again.store, tcn.store)
    # the root node propagates without failure and keeps the hidden solution
    out, failed, _, _, _ = pyoracle.propagate(tcn.store, tcn.props)
    assert not failed and ((out["lb"] <= v) & (v <= out["ub"])).all()


def test_model_statistics_and_random_order():
    from turbo_amd import frontend
    m = frontend.Model.from_file(os.path.join(BENCH, "test_data", "pat2.fzn"))
    st = m.fcn_statistics()
    assert int(st["fcn_variables"]) > 0 and int(st["fcn_constraints"]) == m.tcn().parsed_constraints
    assert "'int_lin_le'" in st["fcn_histogram_symbols"]
    orders = {}
    for seed in (0, 0, 5):
        mm = frontend.Model.from_file(os.path.join(BENCH, "test_data", "pat2.fzn"))
        before = mm.tcn()
        mm.push_eps_strategy("input_order", "min")
        mm.shuffle_strategy(0, seed)
        t = mm.tcn()
        vars0 = t.strat_vars[t.strat_off[0]:t.strat_off[1]].tolist()
        assert sorted(vars0) == sorted(before.strat_vars[before.strat_off[0]:before.strat_off[1]].tolist())
        assert t.strat_var_order[0] == frontend.VAR_ORDERS["input_order"]
        orders.setdefault(seed, []).append(vars0)
    assert orders[0][0] == orders[0][1] and orders[0][0] != orders[5][0]


# ---- bench.py as a launcher (no GPU needed: the ranks fail loudly without one, and the launcher must relay that) ----------

def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline"], env=dict(_clean_env(), WORLD_SIZE="1", RANK="0"),
                       capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_bench_launcher_starts_child_ranks_and_relays_their_failure():
    """`python bench.py --gpus 2` without WORLD_SIZE: the parent spawns two ranks under torch.distributed.run.  Here there is no
    GPU, so every rank stops with the engine's "no CPU fallback" message; the launcher exits non-zero and prints no result line."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--no-cpu-baseline", "--side-steps", "0"],
                       env=_clean_env(), capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode != 0
    assert "no GPU is visible" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
