"""Self-check of the event-driven fixpoint (tuning build of the engine, `make tuning`).

The event-driven fixpoint only re-evaluates the 64-propagator slices that were woken up, drops slices flagged all-entailed,
and evaluates some slices jointly in one pass.  A missed wake-up or a wrong "entailed" flag does not show in a node-level
comparison (tb_propagate starts from scratch) and shows in a tree comparison only as a different tree.  The tuning build can
re-evaluate EVERY propagator with the generic rules after every node that did not fail (tb_config.reserved[0] & 0x1000000)
and report (a) a propagator that could still narrow, (b) a slice flagged all-entailed with a propagator that is not.
This test runs searches in that mode -- one workgroup and a full grid (concurrent waves and workgroups) -- and demands a
clean report.  (It found the r02 bug of the joint channelling run: two slices walking the bounds of one variable towards
each other in the same round emptied it without either of them noticing.)
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
TUNING_LIB = os.path.join(ROOT, "turbo_amd", "lib", "libturbo_hip_tuning.so")

WORKER = r'''
import json, os, sys
sys.path.insert(0, os.environ["TB_ROOT"])
from turbo_amd import capi, frontend, preprocess
out = []
for name, simplify, kw in json.loads(sys.argv[1]):
    path = os.path.join(os.environ["TB_ROOT"], "benchmarks", name)
    tcn = preprocess.load_fzn_simplified(path)[1] if simplify else frontend.load_fzn(path)
    has, best, st = capi.solve(tcn, capi.make_config(fixpoint=2, timeout_ms=120000, debug=0x1000000 | kw.pop("debug", 0), **kw))
    out.append({"name": name, "why": st["why_not_exhaustive"], "slice": st["debug_slice"] - 1, "nodes": st["nodes"]})
print(json.dumps(out))
'''

CASES = [
    ("example_wordpress7_500.fzn", False, dict(or_nodes=1, subproblems_power=0, stop_after_n_nodes=1500)),
    ("example_wordpress7_500.fzn", True, dict(stop_after_n_nodes_total=400000)),
    ("example_wordpress7_500.fzn", False, dict(stop_after_n_nodes_total=300000, threads_per_block=1024)),
    ("accap_a3.fzn", True, dict(stop_after_n_nodes_total=300000, debug=0x100000)),
    ("accap_a3.fzn", False, dict(or_nodes=4, stop_after_n_nodes=3000)),
    ("trains15.fzn", True, dict(stop_after_n_nodes_total=300000)),
    ("trains15.fzn", False, dict(or_nodes=64, stop_after_n_nodes=500)),
    ("test_data/pat7.fzn", False, dict(debug=0x100000)),
    ("test_data/sudoku_opt4.fzn", False, dict()),
    ("test_data/pennies5.fzn", True, dict(debug=0x100000)),
    ("test_data/triangular9.fzn", False, dict(debug=0x100000)),
]


def test_event_fixpoint_passes_its_self_check():
    if not os.path.exists(TUNING_LIB):
        pytest.fail(f"{TUNING_LIB} is missing: run `make tuning` (part of __graft_entry__.build())")
    env = dict(os.environ, TB_ROOT=ROOT, TURBO_HIP_LIB=TUNING_LIB)
    p = subprocess.run([sys.executable, "-c", WORKER, json.dumps(CASES)], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    rows = json.loads(p.stdout.strip().splitlines()[-1])
    assert len(rows) == len(CASES)
    for r in rows:
        assert r["nodes"] > 0
        assert (r["why"] & 0x300) == 0, f"{r['name']}: self-check flags {hex(r['why'] & 0x300)} at slice {r['slice']} (0x100: a propagator can still narrow, 0x200: wrong all-entailed flag)"
