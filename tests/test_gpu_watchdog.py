"""The in-kernel watchdogs (kernels.hpp: deadline_passed; memory_gpu.hpp:174-196 is the reference's host-side wait loop): a pair `x < y`, `y < x` over 2^30 values
converges one value per iteration -- minutes of wave-local iterations -- so a node of it only ends because the wave-local and the block-level watchdogs look at the
deadline.  Every fixpoint and layout, batch propagation and search: the call returns close to its timeout, reports the abort (failed = -1 / not exhaustive) and the
process survives (r04: a watchdog that read its deadline through a wrong address space passed every other test and faulted the GPU the first time its branch ran)."""
import time

import numpy as np
import pytest

from turbo_amd import capi
from turbo_amd.frontend import ITV_DTYPE, PROP_DTYPE, TCN

pytestmark = pytest.mark.gpu
BIG = 2 ** 30


def slow_network(n_pairs=70):
    """n_pairs independent pairs x < y < x (as `0 = (y <= x)`-style reified comparisons with constant truth values) plus Booleans, so that the compact layouts apply."""
    store = [(0, 0), (1, 1), (2, 2)]
    props = []
    for _ in range(n_pairs):
        x = len(store); store.append((0, BIG))
        y = len(store); store.append((0, BIG))
        b = len(store); store.append((0, 1))
        props.append((7, 0, y, x))   # 0 = (y <= x)  i.e. x < y
        props.append((7, 0, x, y))   # 0 = (x <= y)  i.e. y < x
        props.append((7, 1, b, 1))   # b <= 1
    return np.array(store, dtype=ITV_DTYPE), np.array(props, dtype=PROP_DTYPE)


@pytest.mark.parametrize("fixpoint,debug,extra", [(0, 0, {}), (1, 0, {}), (2, 0, {}), (2, 0x100000, {}), (2, 0x100000 | 0x10000000, {}), (1, 0, dict(only_global_memory=1)),
                                                  (2, 0, dict(only_global_memory=1)), (1, 0, dict(threads_per_block=1024)), (1, 0, dict(entailed_prop_removal=1))],
                         ids=["ac1", "wac1", "event", "event_compact", "event_compact16", "wac1_global", "event_global", "wac1_1024", "wac1_rm"])
def test_a_slowly_converging_node_ends_at_the_deadline(fixpoint, debug, extra):
    store, props = slow_network()
    t0 = time.time()
    got, failed, ent, iters, ded, ns = capi.propagate(props, store[None, :], capi.make_config(fixpoint=fixpoint, debug=debug, timeout_ms=400, **extra))
    dt = time.time() - t0
    assert failed[0] == -1, "the node must be reported as aborted, not as failed or consistent"
    assert dt < 20.0
    assert ded[0] > 0


@pytest.mark.parametrize("fixpoint", [1, 2])
def test_a_search_on_it_stops_and_says_so(fixpoint):
    store, props = slow_network()
    tcn = TCN(store=store, props=props, strat_var_order=np.array([0], dtype=np.int32), strat_val_order=np.array([0], dtype=np.int32),
              strat_off=np.array([0, 0], dtype=np.int32), strat_vars=np.zeros(0, dtype=np.int32), obj_var=-1, goal=0, goal_var=-1)
    t0 = time.time()
    has, best, st = capi.solve(tcn, capi.make_config(fixpoint=fixpoint, timeout_ms=500, or_nodes=8))
    assert time.time() - t0 < 30.0
    assert not has and st["exhaustive"] == 0
