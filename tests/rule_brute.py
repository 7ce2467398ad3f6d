"""Brute-force semantics of the eight ternary operators `x = y op z` -- INDEPENDENT of oracle/ and of the engine.

lattice-land/lala-pc (the home of PIR::deduce / PIR::ask) is absent from the reference tree, so the propagator
rules cannot be pinned against the reference's own code.  What CAN be pinned without it is what any correct
implementation of those rules must satisfy, by the definition of the constraints themselves
(FlatZinc: int_plus, int_times, int_div / int_mod truncating, int_min, int_max, int_eq_reif, int_le_reif;
TCN conventions: common_solving.hpp:739-771):

  (i)   soundness   : deduce never removes a satisfying triple of the box;
  (ii)  entailment  : ask => every triple of the box satisfies the constraint;
  (iii) decision    : on a box of three singletons, deduce fails iff the constraint is false, and if it does not
                      fail the propagator is entailed.

This module enumerates boxes over a small universe and counts satisfying triples with summed-area tables, so the
three properties are checked exhaustively (numpy, no loops over triples).  Used by tests/test_propagator_rules.py
for the oracle (CPU) and for the HIP engine (GPU, through tb_propagate).
"""
from __future__ import annotations

import itertools

import numpy as np

NINF, PINF = -(2 ** 31), 2 ** 31 - 1
OPS = ["ADD", "MUL", "TDIV", "TMOD", "MIN", "MAX", "EQ", "LEQ"]
WIN = 24  # window [-WIN, WIN] standing in for the integers: every result of an operator on universe values lies well inside


def tdiv(y, z):
    q = abs(y) // abs(z)
    return q if (y < 0) == (z < 0) else -q


def holds(op: int, x: int, y: int, z: int) -> bool:
    """The constraint itself, on mathematical integers."""
    if op == 0:
        return x == y + z
    if op == 1:
        return x == y * z
    if op == 2:
        return z != 0 and x == tdiv(y, z)
    if op == 3:
        return z != 0 and x == y - z * tdiv(y, z)
    if op == 4:
        return x == min(y, z)
    if op == 5:
        return x == max(y, z)
    if op == 6:
        return (x == 1 and y == z) or (x == 0 and y != z)
    return (x == 1 and y <= z) or (x == 0 and y > z)


def relation_tensor(op: int) -> np.ndarray:
    """R[x + WIN, y + WIN, z + WIN] over the window."""
    v = np.arange(-WIN, WIN + 1, dtype=np.int64)
    X, Y, Z = np.meshgrid(v, v, v, indexing="ij")
    if op == 0:
        return X == Y + Z
    if op == 1:
        return X == Y * Z
    if op in (2, 3):
        Zs = np.where(Z == 0, 1, Z)
        q = np.abs(Y) // np.abs(Zs)
        q = np.where((Y < 0) == (Zs < 0), q, -q)
        r = Y - Zs * q
        return (Z != 0) & (X == (q if op == 2 else r))
    if op == 4:
        return X == np.minimum(Y, Z)
    if op == 5:
        return X == np.maximum(Y, Z)
    if op == 6:
        return ((X == 1) & (Y == Z)) | ((X == 0) & (Y != Z))
    return ((X == 1) & (Y <= Z)) | ((X == 0) & (Y > Z))


def summed_area(R: np.ndarray) -> np.ndarray:
    """S[i, j, k] = number of true cells with index < (i, j, k) (one leading zero plane per axis)."""
    S = np.zeros(tuple(n + 1 for n in R.shape), dtype=np.int64)
    S[1:, 1:, 1:] = R.astype(np.int64).cumsum(0).cumsum(1).cumsum(2)
    return S


def _clip(lb, ub):
    lo = np.clip(lb.astype(np.int64), -WIN, WIN) + WIN
    hi = np.clip(ub.astype(np.int64), -WIN, WIN) + WIN + 1  # exclusive
    return lo, np.maximum(hi, lo)  # an empty interval counts nothing


def count_in_boxes(S: np.ndarray, boxes: np.ndarray) -> np.ndarray:
    """Number of satisfying triples of the window inside each box.  boxes: int64 [N, 6] = xl, xu, yl, yu, zl, zu."""
    x0, x1 = _clip(boxes[:, 0], boxes[:, 1])
    y0, y1 = _clip(boxes[:, 2], boxes[:, 3])
    z0, z1 = _clip(boxes[:, 4], boxes[:, 5])
    empty = (boxes[:, 0] > boxes[:, 1]) | (boxes[:, 2] > boxes[:, 3]) | (boxes[:, 4] > boxes[:, 5])
    c = (S[x1, y1, z1] - S[x0, y1, z1] - S[x1, y0, z1] - S[x1, y1, z0] + S[x0, y0, z1] + S[x0, y1, z0] + S[x1, y0, z0] - S[x0, y0, z0])
    return np.where(empty, 0, c)


def volume_in_window(boxes: np.ndarray) -> np.ndarray:
    x0, x1 = _clip(boxes[:, 0], boxes[:, 1])
    y0, y1 = _clip(boxes[:, 2], boxes[:, 3])
    z0, z1 = _clip(boxes[:, 4], boxes[:, 5])
    return (x1 - x0) * (y1 - y0) * (z1 - z0)


def intervals(lo: int, hi: int, with_inf: bool = True):
    """Every interval with bounds in lo..hi, plus the half-infinite and the infinite ones."""
    out = [(a, b) for a in range(lo, hi + 1) for b in range(a, hi + 1)]
    if with_inf:
        out += [(NINF, b) for b in range(lo, hi + 1)] + [(a, PINF) for a in range(lo, hi + 1)] + [(NINF, PINF)]
    return out


def all_boxes(op: int, lo: int = -3, hi: int = 3) -> np.ndarray:
    """Boxes (X, Y, Z) over the universe; the truth variable of a comparison is a Boolean (common_solving.hpp:743-771)."""
    iv = intervals(lo, hi)
    xs = [(0, 0), (0, 1), (1, 1)] if op >= 6 else iv
    return np.array([x + y + z for x, y, z in itertools.product(xs, iv, iv)], dtype=np.int64)


def check_properties(op: int, boxes: np.ndarray, out: np.ndarray, failed: np.ndarray, entailed: np.ndarray, what: str) -> None:
    """boxes / out: [N, 6] before / after propagation; failed, entailed: bool [N].  Raises AssertionError with a witness."""
    S = summed_area(relation_tensor(op))
    before = count_in_boxes(S, boxes)
    inter = out.copy()
    for k in (0, 2, 4):  # propagation only narrows: intersect anyway so that a widening would show up as a loss below
        inter[:, k] = np.maximum(out[:, k], boxes[:, k])
        inter[:, k + 1] = np.minimum(out[:, k + 1], boxes[:, k + 1])
    after = np.where(failed, 0, count_in_boxes(S, inter))
    bad = np.flatnonzero(before != after)
    assert bad.size == 0, f"{what} {OPS[op]}: a satisfying triple was removed from box {boxes[bad[0]].tolist()} -> {out[bad[0]].tolist()} failed={bool(failed[bad[0]])} ({before[bad[0]]} -> {after[bad[0]]})"
    widened = np.flatnonzero(~failed & ((out[:, 0] < boxes[:, 0]) | (out[:, 1] > boxes[:, 1]) | (out[:, 2] < boxes[:, 2]) | (out[:, 3] > boxes[:, 3]) | (out[:, 4] < boxes[:, 4]) | (out[:, 5] > boxes[:, 5])))
    assert widened.size == 0, f"{what} {OPS[op]}: a domain grew: {boxes[widened[0]].tolist()} -> {out[widened[0]].tolist()}"
    ent = entailed & ~failed
    full = count_in_boxes(S, out) == volume_in_window(out)
    bad = np.flatnonzero(ent & ~full)
    assert bad.size == 0, f"{what} {OPS[op]}: reported entailed but some triple of {out[bad[0]].tolist()} violates it (from {boxes[bad[0]].tolist()})"
    single = (boxes[:, 0] == boxes[:, 1]) & (boxes[:, 2] == boxes[:, 3]) & (boxes[:, 4] == boxes[:, 5])
    bad = np.flatnonzero(single & (failed != (before == 0)))
    assert bad.size == 0, f"{what} {OPS[op]}: singleton box {boxes[bad[0]].tolist()}: failed={bool(failed[bad[0]])} but constraint holds={before[bad[0]] != 0}"
    bad = np.flatnonzero(single & ~failed & ~entailed)
    assert bad.size == 0, f"{what} {OPS[op]}: satisfied singleton box {boxes[bad[0]].tolist()} is not entailed"


def extreme_boxes(op: int) -> np.ndarray:
    """Narrow boxes around the ends of the 32-bit range (saturation, sentinels next to finite values)."""
    vals = [NINF + 1, NINF + 2, -2, -1, 0, 1, 2, PINF - 2, PINF - 1]
    iv = [(a, a) for a in vals] + [(a, b) for a, b in zip(vals, vals[1:]) if b - a == 1] + [(NINF, NINF + 2), (PINF - 2, PINF), (NINF, PINF)]
    xs = [(0, 0), (0, 1), (1, 1)] if op >= 6 else iv
    return np.array([x + y + z for x, y, z in itertools.product(xs, iv, iv)], dtype=np.int64)


def check_extreme(op: int, boxes: np.ndarray, out: np.ndarray, failed: np.ndarray, entailed: np.ndarray, what: str) -> None:
    """Python-integer brute force on the narrow boxes of extreme_boxes (finite values = NINF+1 .. PINF-1)."""
    def values(lb, ub):
        lb, ub = max(int(lb), NINF + 1), min(int(ub), PINF - 1)
        if ub - lb > 8:  # a (half-)infinite interval: sample both ends and the middle
            cand = list(range(lb, lb + 3)) + list(range(ub - 2, ub + 1)) + [-2, -1, 0, 1, 2]
            return sorted({v for v in cand if lb <= v <= ub})
        return list(range(lb, ub + 1))
    for b, o, f, e in zip(boxes.tolist(), out.tolist(), failed.tolist(), entailed.tolist()):
        sat = [(x, y, z) for x in values(b[0], b[1]) for y in values(b[2], b[3]) for z in values(b[4], b[5]) if holds(op, x, y, z)]
        if f:
            assert not sat, f"{what} {OPS[op]}: box {b} failed but {sat[0]} satisfies it"
            continue
        for (x, y, z) in sat:
            assert o[0] <= x <= o[1] and o[2] <= y <= o[3] and o[4] <= z <= o[5], f"{what} {OPS[op]}: {b} -> {o} lost the satisfying triple {(x, y, z)}"
        if e:
            exhaustive = all(u - l <= 8 for l, u in ((o[0], o[1]), (o[2], o[3]), (o[4], o[5])))
            allt = [(x, y, z) for x in values(o[0], o[1]) for y in values(o[2], o[3]) for z in values(o[4], o[5])]
            viol = [t for t in allt if not holds(op, *t)]
            assert not viol, f"{what} {OPS[op]}: {o} reported entailed but {viol[0]} violates it" + ("" if exhaustive else " (sampled)")
        if b[0] == b[1] and b[2] == b[3] and b[4] == b[5] and NINF < b[0] < PINF and NINF < b[2] < PINF and NINF < b[4] < PINF:
            assert e == bool(sat), f"{what} {OPS[op]}: singleton box {b}: entailed={e}, holds={bool(sat)}"
