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
