"""Oracle-independent pin of the propagator rules.

lala-pc is absent, so `deduce` / `ask` are restated both in oracle/oracle.c and in the HIP engine; comparing one with
the other cannot catch a shared misunderstanding.  These tests check each of them against the DEFINITION of the eight
constraints by brute force (tests/rule_brute.py): soundness, entailment and decision on singletons, exhaustively
over all boxes with bounds in -3..3 plus the (half-)infinite ones, and on narrow boxes at the ends of the 32-bit range.
"""
import ctypes as C

import numpy as np
import pytest

import rule_brute as rb
from oracle import pyoracle


def boxes_to_stores(boxes: np.ndarray) -> np.ndarray:
    st = np.zeros((boxes.shape[0], 3), dtype=pyoracle.ITV)
    for k in range(3):
        st["lb"][:, k] = boxes[:, 2 * k]
        st["ub"][:, k] = boxes[:, 2 * k + 1]
    return st


def stores_to_boxes(st: np.ndarray) -> np.ndarray:
    out = np.zeros((st.shape[0], 6), dtype=np.int64)
    for k in range(3):
        out[:, 2 * k] = st["lb"][:, k]
        out[:, 2 * k + 1] = st["ub"][:, k]
    return out


def oracle_fixpoints(op: int, boxes: np.ndarray):
    """Fixpoint of the single propagator `v0 = v1 op v2` on every box, with the oracle."""
    L = pyoracle.lib()
    st = boxes_to_stores(boxes)
    prop = np.array([(op, 0, 1, 2)], dtype=pyoracle.PROP)
    failed = np.zeros(boxes.shape[0], dtype=bool)
    ent = np.zeros(boxes.shape[0], dtype=bool)
    e = C.c_int(0)
    base, pp = st.ctypes.data, prop.ctypes.data
    for i in range(boxes.shape[0]):
        failed[i] = L.orc_propagate(3, base + 24 * i, 1, pp, None, None, C.byref(e)) != 0
        ent[i] = e.value != 0
    return stores_to_boxes(st), failed, ent


def oracle_single_step(op: int, boxes: np.ndarray):
    """One application of orc_deduce, and orc_ask on the INPUT box (the two entry points the fixpoint is made of)."""
    L = pyoracle.lib()
    st = boxes_to_stores(boxes)
    ask_in = np.zeros(boxes.shape[0], dtype=bool)
    failed = np.zeros(boxes.shape[0], dtype=bool)
    prop = np.array([(op, 0, 1, 2)], dtype=pyoracle.PROP)
    f = C.c_int(0)
    base, pp = st.ctypes.data, prop.ctypes.data
    for i in range(boxes.shape[0]):
        ask_in[i] = L.orc_ask(pp, base + 24 * i) != 0
        f.value = 0
        L.orc_deduce(pp, base + 24 * i, C.byref(f))
        failed[i] = f.value != 0
    return stores_to_boxes(st), failed, ask_in


@pytest.mark.parametrize("op", range(8), ids=rb.OPS)
def test_oracle_rules_are_sound_and_decide(op):
    boxes = rb.all_boxes(op)
    out, failed, ent = oracle_fixpoints(op, boxes)
    rb.check_properties(op, boxes, out, failed, ent, "oracle fixpoint")


@pytest.mark.parametrize("op", range(8), ids=rb.OPS)
def test_oracle_single_deduce_and_ask(op):
    boxes = rb.all_boxes(op, -2, 2)
    out, failed, ask_in = oracle_single_step(op, boxes)
    # one deduce step: soundness only (entailment / decision are properties of the fixpoint) ...
    S = rb.summed_area(rb.relation_tensor(op))
    before = rb.count_in_boxes(S, boxes)
    inter = out.copy()
    for k in (0, 2, 4):
        inter[:, k] = np.maximum(out[:, k], boxes[:, k])
        inter[:, k + 1] = np.minimum(out[:, k + 1], boxes[:, k + 1])
    after = np.where(failed, 0, rb.count_in_boxes(S, inter))
    bad = np.flatnonzero(before != after)
    assert bad.size == 0, f"orc_deduce {rb.OPS[op]}: {boxes[bad[0]].tolist()} -> {out[bad[0]].tolist()} lost a satisfying triple"
    # ... and ask on the input box implies that the whole box satisfies the constraint
    full = before == rb.volume_in_window(boxes)
    bad = np.flatnonzero(ask_in & ~full)
    assert bad.size == 0, f"orc_ask {rb.OPS[op]}: {boxes[bad[0]].tolist()} is not entailed"


@pytest.mark.parametrize("op", range(8), ids=rb.OPS)
def test_oracle_rules_at_the_ends_of_the_32_bit_range(op):
    boxes = rb.extreme_boxes(op)
    out, failed, ent = oracle_fixpoints(op, boxes)
    rb.check_extreme(op, boxes, out, failed, ent, "oracle")


def test_brute_force_semantics_self_check():
    """The checker's two statements of the constraints (scalar `holds`, vectorised `relation_tensor`) agree."""
    rng = np.random.default_rng(0)
    for op in range(8):
        R = rb.relation_tensor(op)
        for _ in range(400):
            x, y, z = (int(v) for v in rng.integers(-rb.WIN, rb.WIN + 1, size=3))
            if op >= 6:
                x = int(rng.integers(0, 2))
            assert bool(R[x + rb.WIN, y + rb.WIN, z + rb.WIN]) == rb.holds(op, x, y, z), (op, x, y, z)
    assert rb.tdiv(-7, 2) == -3 and rb.tdiv(7, -2) == -3 and rb.tdiv(-7, -2) == 3  # FlatZinc int_div truncates


# ---- GPU twin: the same boxes through tb_propagate (one store per box, one workgroup per store) ---------------------

def engine_fixpoints(op: int, boxes: np.ndarray, **cfg):
    from turbo_amd import capi
    from turbo_amd.frontend import PROP_DTYPE
    props = np.array([(op, 0, 1, 2)], dtype=PROP_DTYPE)
    st = boxes_to_stores(boxes)
    got, failed, ent, _, _, _ = capi.propagate(props, st, capi.make_config(timeout_ms=60000, **cfg))
    return stores_to_boxes(got), failed != 0, ent != 0


def _split_by_truth(op, boxes):
    """Comparisons: batches with the same truth domain, so that the engine's constant-truth classes (y <= z, y > z, y = z,
    y != z: a class is chosen when the variable is the same singleton in every store of the batch) are exercised too."""
    if op < 6:
        return [boxes]
    keys = boxes[:, 0] * 2 + boxes[:, 1]
    return [boxes[keys == k] for k in np.unique(keys)] + [boxes]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["wac1", "ac1", "event", "event_compact", "wac1_globalmem"])
@pytest.mark.parametrize("op", range(8), ids=rb.OPS)
def test_engine_rules_are_sound_and_decide(op, mode):
    cfg = {"wac1": dict(fixpoint=1), "ac1": dict(fixpoint=0), "event": dict(fixpoint=2), "event_compact": dict(fixpoint=2, debug=0x100000),
           "wac1_globalmem": dict(fixpoint=1, only_global_memory=1)}[mode]
    for part in _split_by_truth(op, rb.all_boxes(op)):
        out, failed, ent = engine_fixpoints(op, part, **cfg)
        rb.check_properties(op, part, out, failed, ent, f"engine[{mode}]")


@pytest.mark.gpu
@pytest.mark.parametrize("op", range(8), ids=rb.OPS)
def test_engine_rules_at_the_ends_of_the_32_bit_range(op):
    for part in _split_by_truth(op, rb.extreme_boxes(op)):
        for cfg in (dict(fixpoint=1), dict(fixpoint=2)):
            out, failed, ent = engine_fixpoints(op, part, **cfg)
            rb.check_extreme(op, part, out, failed, ent, "engine")
