"""pytest configuration: `gpu` marker, repo path, shared helpers."""
import csv
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BENCH = os.path.join(ROOT, "benchmarks")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")
    config.addinivalue_line("markers", "soak: the part of a wide instance x mode sweep that `pytest -m gpu` leaves out (scripts/gpu_full.sh runs it: `pytest -m \"gpu and soak\"`)")


# Wide sweeps: several GPU families run the reference's ~30 known-answer instances under every kernel mode (6-8 of them) -- 200+ searches a family, most of
# them the same code on one more small instance; the suite had grown to 559 s against the driver's 1200 s limit.  `pytest -m gpu` now runs, for every such family,
#   * EVERY mode on a representative set of instances (REPRESENTATIVE: optimisation / satisfaction-like / unconstrained / reified / regression instances), and
#   * ONE mode, rotating, on every other instance -- so each instance still meets each family, and each mode still meets a dozen instances;
# the rest of the cross product carries the `soak` marker and is deselected unless the -m expression names it (or TB_SOAK=1): scripts/gpu_full.sh.
REPRESENTATIVE = {"test_data/sudoku_opt4.fzn", "test_data/pat2.fzn", "test_data/pat7.fzn", "test_data/pennies5.fzn", "test_data/bug4.fzn", "test_data/pat11.fzn",
                  "test_data/reified_in.fzn", "test_data/minimize_unconstrained.fzn"}
WIDE_MIN_INSTANCES = 20


def pytest_collection_modifyitems(config, items):
    by_fn = {}
    for it in items:
        cs = getattr(it, "callspec", None)
        if cs is None or "rel" not in cs.params or it.get_closest_marker("gpu") is None:
            continue
        by_fn.setdefault((it.module.__name__, it.originalname), []).append(it)
    soak = []
    for fn_items in by_fn.values():
        rels, modes = [], []
        for it in fn_items:
            rel = str(it.callspec.params["rel"]).replace("benchmarks/", "")
            mode = tuple(sorted((k, repr(v)) for k, v in it.callspec.params.items() if k not in ("rel", "expected")))
            if rel not in rels:
                rels.append(rel)
            if mode not in modes:
                modes.append(mode)
        if len(rels) < WIDE_MIN_INSTANCES or len(modes) < 2:
            continue
        for it in fn_items:
            rel = str(it.callspec.params["rel"]).replace("benchmarks/", "")
            mode = tuple(sorted((k, repr(v)) for k, v in it.callspec.params.items() if k not in ("rel", "expected")))
            if rel in REPRESENTATIVE or modes.index(mode) == rels.index(rel) % len(modes):
                continue
            it.add_marker(pytest.mark.soak)
            soak.append(it)
    # the 5 layouts x 5 variable orders x 4 value orders of tests/test_gpu_orders.py: every order on the first two layouts, every second order on the others
    for it in items:
        cs = getattr(it, "callspec", None)
        if cs is None or it.originalname != "test_every_order_on_class_pure_networks":
            continue
        layout_index = [i for i, x in enumerate(it.module.LAYOUTS) if x == (cs.params["fixpoint"], cs.params["debug"])]
        if layout_index and layout_index[0] >= 2 and (cs.params["var_order"] + cs.params["val_order"] + layout_index[0]) % 2 == 1:
            it.add_marker(pytest.mark.soak)
            soak.append(it)
    wanted = "soak" in (config.getoption("-m") or "") or os.environ.get("TB_SOAK") == "1"
    if soak and not wanted:
        gone = set(id(i) for i in soak)
        config.hook.pytest_deselected(items=soak)
        items[:] = [i for i in items if id(i) not in gone]


def known_answers():
    """(relative path, expected objective) rows of the reference's benchmarks/test_list.csv (FlatZinc rows only)."""
    rows = []
    with open(os.path.join(BENCH, "test_list.csv")) as f:
        for path, exp in csv.reader(f):
            if path.endswith(".fzn"):
                rows.append((path.replace("benchmarks/", ""), int(exp)))
    return rows


# instances the sequential oracle cannot finish within a CPU-suite budget (proved only on the GPU)
SLOW_FOR_ORACLE = {"test_data/triangular9.fzn", "test_data/pat12.fzn", "test_data/pat13.fzn"}


@pytest.fixture(scope="session")
def bench_dir():
    return BENCH
