"""Full-grid parity of the PRODUCTION search kernels at BASELINE size.

The headline-tree tests pin one workgroup against the oracle, node for node; a full grid cannot be pinned that way -- which workgroup gets which
subproblem, and which incumbent it sees when, depends on timing.  What does not depend on timing is the store under a given node: it is the
fixpoint of root + the decisions on the path + the objective bound in force, whatever snapshots and older bounds the workgroup went through.
So: run bench.py's exact configuration (the engine's own plan, asserted below) with a node budget, ask sampled workgroups where they stood when
the budget ended the search (tb_session_debug_path: subproblem, decisions with the bound in force at each, last bound) and let the ORACLE
replay each path from the root (oracle.c: orc_replay_path -- root, dive along the bits of the subproblem index, the decisions, the last node).
Demanded, per sampled workgroup: every recorded decision is the one the oracle's variable selection takes on the replayed store (variable and
both children), the failed flag of the last node agrees, and the store under it is the oracle's, bit for bit.
"""
import os

import numpy as np
import pytest

from conftest import BENCH
from oracle import pyoracle
from turbo_amd import capi, preprocess

pytestmark = pytest.mark.gpu
KEEP = 0x800000  # tb_config.reserved[0]: keep every workgroup's last store and path

# instance -> (node budget, the plan bench.py's line reports on an MI355X: workgroups, threads, kernel_event, kernel_opt, memory kind)
CASES = {
    "example_wordpress7_500.fzn": (2_000_000, dict(num_blocks=3584, threads_per_block=128, kernel_event=1, kernel_opt=1, mem_kind=1)),
    "accap_a3.fzn": (1_500_000, dict(num_blocks=3584, threads_per_block=128, kernel_event=1, kernel_opt=0, mem_kind=1)),
    "trains15.fzn": (800_000, dict(num_blocks=3072, threads_per_block=128, kernel_event=1, kernel_opt=4, mem_kind=1)),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_sampled_workgroups_of_the_full_grid_stand_on_the_oracles_stores(name):
    budget, want = CASES[name]
    _, tcn, _ = preprocess.load_fzn_simplified(os.path.join(BENCH, name))
    s = capi.Session(tcn, capi.make_config(fixpoint=2, stop_after_n_nodes_total=budget, timeout_ms=300000, debug=KEEP))
    plan = s.plan()
    info = capi.device_info(0)
    if info["compute_units"] == 256 and info["lds_bytes_per_cu"] == 160 * 1024:  # an MI355X: the plan must be the one BENCH prints
        for k, v in want.items():
            assert plan[k] == v, (name, k, plan)
    s.start()
    while not s.poll()[1]:
        pass
    has, best, st = s.finish()
    assert st["nodes"] >= budget and not st["exhaustive"], "the budget must end the search"
    B = plan["num_blocks"]
    sample = sorted(set(int(x) for x in np.linspace(0, B - 1, 160)))
    checked = mid_dive = failed_nodes = deepest = 0
    for wg in sample:
        hdr, dec = s.debug_path(wg)
        if not hdr["had_work"] or hdr["depth"] != hdr["decisions"]:
            continue  # left for lack of work, or deeper than the decisions handed out
        last = s.debug_last_store(wg)
        store, failed, mismatch = pyoracle.replay_path(tcn, plan["subproblems_power"], hdr, dec)
        assert mismatch == -1, f"{name} workgroup {wg}: the oracle does not take decision {mismatch} of {hdr}"
        assert failed == bool(hdr["last_node_failed"]), f"{name} workgroup {wg}: failed flag of the last node ({hdr})"
        if not failed:
            np.testing.assert_array_equal(store, last, err_msg=f"{name} workgroup {wg} {hdr}")
        checked += 1
        mid_dive += hdr["dive_levels_left"] > 0
        failed_nodes += failed
        deepest = max(deepest, hdr["depth"])
    s.close()
    assert checked >= 64, f"{name}: only {checked} of the {len(sample)} sampled workgroups were still searching"
    assert checked - failed_nodes >= 24, f"{name}: only {checked - failed_nodes} stores were compared (the other last nodes had failed)"
    print(f"{name}: {checked} workgroups replayed ({mid_dive} stopped in their dive, {failed_nodes} on a failed node, deepest path {deepest} decisions), "
          f"2^{plan['subproblems_power']} subproblems, {st['nodes']} nodes")
