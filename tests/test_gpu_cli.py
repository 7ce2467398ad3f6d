"""GPU tests of the `turbo` executable: the reference's regression script (test_turbo.sh) restated."""
import os
import re
import subprocess

import pytest

from conftest import BENCH, ROOT, known_answers

pytestmark = pytest.mark.gpu
TURBO = os.path.join(ROOT, "turbo_amd", "bin", "turbo")


@pytest.mark.parametrize("fp", ["wac1", "event"])
@pytest.mark.parametrize("simplify_flag", ["", "-disable_simplify"], ids=["simplify", "disable_simplify"])  # test_turbo.sh:8
@pytest.mark.parametrize("rel,expected", known_answers())
def test_regression_script_contract(rel, expected, simplify_flag, fp):
    # test_turbo.sh:34-44: -eps_var_order input_order -eps_value_order min -arch barebones [-disable_simplify] -s -t 60000
    r = subprocess.run([TURBO, "-eps_var_order", "input_order", "-eps_value_order", "min", "-arch", "barebones", *([simplify_flag] if simplify_flag else []),
                        "-fp", fp, "-s", "-t", "60000", os.path.join(BENCH, rel)], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr
    m = re.search(r"objective=(-?\d+)", r.stdout)            # test_turbo.sh:47
    t = re.search(r"solveTime=([0-9.]+)", r.stdout)          # test_turbo.sh:48
    assert m and t, r.stdout[-2000:]
    assert int(m.group(1)) == expected
    assert float(t.group(1)) < 60.0
    assert "==========" in r.stdout and "----------" in r.stdout


@pytest.mark.parametrize("rel,expected", [r for r in known_answers() if r[0].split("/")[-1] in
                                          ("pennies5.fzn", "bug2.fzn", "pat8.fzn", "sudoku_opt_p0.fzn", "maximize_unconstrained.fzn")])
def test_simplifier_can_be_disabled(rel, expected):
    outs = {}
    for extra in ([], ["-disable_simplify"]):
        r = subprocess.run([TURBO, "-arch", "barebones", "-s", "-t", "60000", *extra, os.path.join(BENCH, rel)],
                           capture_output=True, text=True, timeout=180)
        assert r.returncode == 0, r.stderr
        assert int(re.search(r"objective=(-?\d+)", r.stdout).group(1)) == expected
        assert "==========" in r.stdout
        outs[bool(extra)] = r.stdout
    assert "preprocessed_tcn_variables=" in outs[False] and "preprocessing_eliminated_variables=" in outs[False]
    assert "preprocessed_tcn_variables=" not in outs[True]
    tcn_v = int(re.search(r"mzn-stat: tcn_variables=(\d+)", outs[False]).group(1))
    pre_v = int(re.search(r"preprocessed_tcn_variables=(\d+)", outs[False]).group(1))
    assert pre_v <= tcn_v
    assert int(re.search(r"mzn-stat: variables=(\d+)", outs[False]).group(1)) == pre_v
    assert int(re.search(r"mzn-stat: variables=(\d+)", outs[True]).group(1)) == tcn_v


def test_output_format_and_gpu_arch_alias():
    r = subprocess.run([TURBO, "-arch", "gpu", "-s", "-t", "30000", os.path.join(BENCH, "test_data", "sudoku_opt2.fzn")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "objective = -2;" in r.stdout and "x = array2d(1..2, 1..2, [" in r.stdout
    for key in ("nodes", "failures", "variables", "propagators", "peakDepth", "initTime", "solveTime", "num_solutions",
                "eps_num_subproblems", "eps_solved_subproblems", "eps_skipped_subproblems", "num_blocks_done",
                "fixpoint_iterations", "num_deductions", "memory_configuration"):
        assert f"%%%mzn-stat: {key}=" in r.stdout, key


@pytest.mark.parametrize("seed", [0, 7])
def test_random_eps_variable_order(seed):
    # -eps_var_order random = input order over the strategy's variables shuffled with mt19937(seed) (common_solving.hpp:632)
    rel, expected = "test_data/pat8.fzn", 11
    r = subprocess.run([TURBO, "-eps_var_order", "random", "-eps_value_order", "min", "-seed", str(seed), "-s", "-t", "60000",
                        os.path.join(BENCH, rel)], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr
    assert int(re.search(r"objective=(-?\d+)", r.stdout).group(1)) == expected and "==========" in r.stdout
    assert f"%%%mzn-stat: seed={seed}" in r.stdout


def test_network_analysis_statistics():
    path = os.path.join(BENCH, "test_data", "pat2.fzn")
    r = subprocess.run([TURBO, "-s", "-t", "60000", path], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr
    for key in ("fcn_variables", "fcn_constraints", "fcn_var_occurrences", "fcn_histogram_symbols",
                "tcn_variables", "tcn_constraints", "tcn_assigned_variables", "tcn_unbounded_variables", "tcn_histogram_symbols",
                "tcn_histogram_reified_predicates", "tcn_histogram_unassigned_vars_degree", "tcn_histogram_vars_dom_size",
                "preprocessed_tcn_variables", "preprocessed_tcn_histogram_symbols"):
        assert f"%%%mzn-stat: {key}=" in r.stdout, key
    # the histogram of symbols accounts for every ternary constraint
    hist = re.search(r'mzn-stat: tcn_histogram_symbols="\{([^}]*)\}"', r.stdout).group(1)
    total = sum(int(x.split(":")[1]) for x in hist.split(","))
    assert total == int(re.search(r"mzn-stat: tcn_constraints=(\d+)", r.stdout).group(1))
    q = subprocess.run([TURBO, "-s", "-disable_network_analysis", "-t", "60000", path], capture_output=True, text=True, timeout=180)
    assert q.returncode == 0 and "tcn_variables=" in q.stdout and "histogram" not in q.stdout and "fcn_variables" not in q.stdout


def _stat_dict(out, key):
    txt = re.search(r'mzn-stat: %s="\{([^}]*)\}"' % key, out).group(1)
    d = {}
    for item in filter(None, (x.strip() for x in txt.split(","))):
        k, v = item.rsplit(":", 1)
        k = k.strip()
        d[k.strip("'") if k.startswith("'") else int(k)] = int(v)
    return d


@pytest.mark.parametrize("rel", ["test_data/pat2.fzn", "test_data/sudoku_opt4.fzn", "example_wordpress7_500.fzn"])
def test_network_analysis_histogram_values(rel):
    """Values, not only keys, of analyze_tcn (common_solving.hpp:728-826), recomputed here with numpy from the lowered
    network: symbols (a comparison with truth 0 is its negation, an undecided one is reified too), degrees split by
    assigned / unassigned, domain sizes of the bounded variables."""
    from collections import Counter
    import numpy as np
    from turbo_amd import frontend
    path = os.path.join(BENCH, rel)
    r = subprocess.run([TURBO, "-s", "-t", "3000", path], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr
    tcn = frontend.load_fzn(path)
    lb, ub = tcn.store["lb"].astype(np.int64), tcn.store["ub"].astype(np.int64)
    pr = tcn.props
    sym = ["+", "*", "/", "%", "min", "max", "=", "<="]
    ops, reified = Counter(), Counter()
    for op, x in zip(pr["op"].tolist(), pr["x"].tolist()):
        if op >= 6:
            if lb[x] == ub[x]:
                ops[("!=" if op == 6 else ">") if lb[x] == 0 else sym[op]] += 1
            else:
                ops[sym[op]] += 1
                reified[sym[op]] += 1
        else:
            ops[sym[op]] += 1
    occ = np.bincount(np.concatenate([pr["x"], pr["y"], pr["z"]]), minlength=len(lb))
    inf = (lb == -2 ** 31) | (ub == 2 ** 31 - 1)
    assigned = ~inf & (lb == ub)
    assert _stat_dict(r.stdout, "tcn_histogram_symbols") == dict(ops)
    assert _stat_dict(r.stdout, "tcn_histogram_reified_predicates") == dict(reified)
    assert _stat_dict(r.stdout, "tcn_histogram_assigned_vars_degree") == dict(Counter(occ[assigned].tolist()))
    assert _stat_dict(r.stdout, "tcn_histogram_unassigned_vars_degree") == dict(Counter(occ[~assigned].tolist()))
    assert _stat_dict(r.stdout, "tcn_histogram_vars_dom_size") == dict(Counter((ub - lb + 1)[~inf].tolist()))
    g = lambda k: int(re.search(r"mzn-stat: %s=(\d+)" % k, r.stdout).group(1))
    assert g("tcn_variables") == len(lb) and g("tcn_constraints") == len(pr)
    assert g("tcn_assigned_variables") == int(assigned.sum()) and g("tcn_unbounded_variables") == int(inf.sum())
    assert g("tcn_assigned_var_occurrences") == int(occ[assigned].sum())
    assert g("tcn_unassigned_var_occurrences") == int(occ[~assigned].sum())


@pytest.mark.parametrize("fp", ["wac1", "ac1"])
@pytest.mark.parametrize("rel,expected", [("test_data/pat7.fzn", None), ("test_data/sudoku_opt4.fzn", None), ("test_data/pennies5.fzn", None)])
def test_sweeps_with_entailed_removal_prove_the_same_optimum(rel, expected, fp):
    """`-entailed_removal` with the sweeping fixpoints (FixpointSubsetGPU of gpu_dive_and_solve.hpp:334, off by default in the
    reference): same optimum, proved, no more propagator evaluations than the plain sweeps."""
    from conftest import known_answers
    want = dict(known_answers())[rel]
    runs = {}
    for extra in ([], ["-entailed_removal"]):
        r = subprocess.run([TURBO, "-s", "-t", "60000", "-fp", fp, "-or", "1", "-sub", "0"] + extra + [os.path.join(BENCH, rel)], capture_output=True, text=True, timeout=180)
        assert r.returncode == 0, r.stderr
        assert "==========" in r.stdout and int(re.search(r"objective=(-?\d+)", r.stdout).group(1)) == want
        runs[bool(extra)] = int(re.search(r"mzn-stat: num_deductions=(\d+)", r.stdout).group(1))
        assert ('entailed_prop_removal="by_slice_entailment"' in r.stdout) == bool(extra)
    assert runs[True] <= runs[False] * 1.1  # (slices are counted whole with removal: a small network without entailed slices may count a few idle lanes more)


def test_timeout_is_honoured_and_reported():
    r = subprocess.run([TURBO, "-s", "-t", "1500", os.path.join(BENCH, "example_wordpress7_500.fzn")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    t = float(re.search(r"solveTime=([0-9.]+)", r.stdout).group(1))
    assert t < 10.0
    assert "==========" not in r.stdout  # not exhaustive


def test_bench_contract_line():
    """bench.py prints ONE JSON line with the keys the driver and the judge read."""
    import json
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--side-steps", "1", "--other-steps", "1",
                        "--reference-seconds", "2", "--sharded-reps", "1", "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline", "nodes_per_sec", "wac1_mode", "balance"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "propagations/s" and d["dtype"] == "int32" and "workload" in d["config"] and "model" not in d["config"]
    roof = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    # the store of the headline workload is LDS resident: the fraction is priced against the level that serves the bytes, never above 1
    assert roof["bound"] == "lds" and 0 < roof["frac"] <= 1 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert 0 < roof["records_from_l2"]["frac"] <= 1
    # r06: every level priced against its own peak, and the rows of stores in global memory name the level the counters show serving them
    assert set(roof["levels"]) == {"lds", "l2", "hbm"} and 0 < roof["levels"]["lds"]["frac"] <= 1 and 0 < roof["levels"]["l2"]["frac"] <= 1
    syn = {o["fixpoint"]: o["roofline"] for o in d["other_workloads"] if o["workload"].startswith("synthetic")}
    assert set(syn) == {"event", "wac1", "ac1"}
    for fp, ro in syn.items():
        assert ro["bound"] in ("l2", "hbm") and 0 < ro["frac"] <= 1 and "bound_chosen_by" in ro, (fp, ro)
    assert syn["wac1"]["bound"] == "l2" and syn["event"]["bound"] == "hbm", {k: v["bound"] for k, v in syn.items()}  # (profiles/r0x_counters.json: L2 hit rate 0.93 against 0.43)
    # r06: the sharded search itself is in the default line -- a whole search of fixed total work (what SCALE_r*.json needs at N > 1)
    sh = d["sharded_search"]
    assert sh["exhaustive"] == 1 and sh["has_solution"] == 0 and sh["every_subproblem_accounted_once"] and sh["eps_solved"] + sh["eps_skipped"] == 1 << sh["subproblems_power"]
    assert 0 < sh["seconds"] < 30 and sh["scaling"] == "strong" and len(sh["per_rank"]) == 1
    assert d["cpu_baseline"]["leaf_rule"] == d["config"]["leaf_rule"] == "barebones"
    cpu = d["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["value"] > 0 and "sample" in cpu and "dfs_sample" in cpu
    assert d["value"] > 10 * cpu["value"]  # north star: >= 10x the CPU propagation rate
    assert d["wac1_mode"]["propagations_per_sec"] > 10 * cpu["value"]


@pytest.mark.parametrize("fp", ["wac1", "event"])
@pytest.mark.parametrize("simplify_flag", [[], ["-disable_simplify"]], ids=["simplify", "disable_simplify"])
def test_unsatisfiable_root_is_reported(tmp_path, fp, simplify_flag):
    """An inconsistent root (found by the GPU root fixpoint, not by the parser): the CLI takes `failed` from tb_propagate
    explicitly (turbo_main.cpp: simplify_network) and prints the reference's separator (statistics.hpp:394-412)."""
    f = tmp_path / "unsat_root.fzn"
    f.write_text("var 1..5: x;\nvar 1..5: y;\nvar 1..5: z;\nconstraint int_lt(x,y);\nconstraint int_lt(y,z);\nconstraint int_lt(z,x);\nsolve satisfy;\n")
    r = subprocess.run([TURBO, "-arch", "gpu", "-fp", fp, "-s", "-t", "30000", *simplify_flag, str(f)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "=====UNSATISFIABLE=====" in r.stdout and "----------" not in r.stdout


def test_two_gpus_flag_runs_two_linked_sessions():
    """`-gpus 2` on a one-GPU box is refused cleanly (device ordinal out of range), never a crash or a silent single-GPU run."""
    r = subprocess.run([TURBO, "-arch", "gpu", "-gpus", "2", "-s", "-t", "30000", os.path.join(BENCH, "test_data", "sudoku_opt2.fzn")],
                       capture_output=True, text=True, timeout=120)
    from turbo_amd import capi
    if capi.lib().tb_device_count() >= 2:
        assert r.returncode == 0 and "objective = -2;" in r.stdout
        assert "%%%mzn-stat: eps_stolen_subproblems=" in r.stdout
    else:
        assert r.returncode != 0 and "device ordinal out of range" in r.stderr


@pytest.mark.parametrize("rel,expected", [r for r in known_answers() if r[0].split("/")[-1] in ("pat7.fzn", "sudoku_opt4.fzn", "pennies5.fzn", "bug4.fzn")])
@pytest.mark.parametrize("fp", ["wac1", "event"])
def test_two_ranks_on_one_device_through_the_cli(rel, expected, fp):
    """`turbo -devices 0,0`: the C++ host's multi-GPU path (solve_sessions: link, arm, start, relay, merge) with both ranks on
    the only GPU of the box."""
    r = subprocess.run([TURBO, "-arch", "gpu", "-devices", "0,0", "-or", "64", "-sub", "10", "-fp", fp, "-s", "-t", "60000", os.path.join(BENCH, rel)],
                       capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr
    assert int(re.search(r"objective=(-?\d+)", r.stdout).group(1)) == expected
    assert "==========" in r.stdout
    solved = int(re.search(r"eps_solved_subproblems=(\d+)", r.stdout).group(1))
    skipped = int(re.search(r"eps_skipped_subproblems=(\d+)", r.stdout).group(1))
    assert solved + skipped == 1024
    assert int(re.search(r"mzn-stat: num_blocks=(\d+)", r.stdout).group(1)) == 128
