"""ctypes loader of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under turbo_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

ITV = np.dtype([("lb", np.int32), ("ub", np.int32)])
PROP = np.dtype([("op", np.int32), ("x", np.int32), ("y", np.int32), ("z", np.int32)])


class OrcConfig(C.Structure):
    _fields_ = [("subproblems_power", C.c_int32), ("has_eps_strategy", C.c_int32),
                ("use_fixed_bound", C.c_int32), ("fixed_bound", C.c_int32),
                ("stop_after_n_nodes", C.c_uint64), ("stop_after_n_solutions", C.c_uint64),
                ("timeout_ms", C.c_uint64), ("leaf_requires_assignment", C.c_int32), ("reserved", C.c_int32)]


class OrcStats(C.Structure):
    _fields_ = [("nodes", C.c_uint64), ("fails", C.c_uint64), ("solutions", C.c_uint64),
                ("fixpoint_iterations", C.c_uint64), ("num_deductions", C.c_uint64),
                ("eps_num_subproblems", C.c_uint64), ("eps_solved_subproblems", C.c_uint64),
                ("eps_skipped_subproblems", C.c_uint64),
                ("depth_max", C.c_int32), ("exhaustive", C.c_int32),
                ("best_bound", C.c_int32), ("best_subproblem", C.c_int32),
                ("solve_seconds", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "oracle.c")):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True, capture_output=True)
    return _LIB_PATH


def build_native(out_dir: str) -> str | None:
    """bench.py's cpu_baseline leg: the same oracle.c compiled `-O3 -march=native` ON THE MACHINE THAT TIMES IT
    (cpu-release flags; the prebuilt liboracle.so is portable -O3).  Returns the path, or None if gcc fails."""
    out = os.path.join(out_dir, "liboracle_native.so")
    try:
        subprocess.run(["gcc", "-O3", "-march=native", "-std=c11", "-fPIC", "-shared", "-o", out, os.path.join(_HERE, "oracle.c")],
                       check=True, capture_output=True, timeout=120)
        return out
    except Exception:
        return None


def use_library(path: str) -> None:
    """Load the oracle from another build of oracle.c (see build_native)."""
    global _lib, _LIB_PATH
    _LIB_PATH, _lib = path, None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if _LIB_PATH == os.path.join(_HERE, "liboracle.so"):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_deduce.restype = C.c_int
        L.orc_deduce.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.orc_ask.restype = C.c_int
        L.orc_ask.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_propagate.restype = C.c_int
        L.orc_propagate.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                    C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        L.orc_solve.restype = C.c_int
        L.orc_solve.argtypes = [C.POINTER(OrcConfig), C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_int32, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(OrcStats)]
        _lib = L
    return _lib


def propagate(store: np.ndarray, props: np.ndarray):
    """One node.  Returns (store_out, failed, all_entailed, iterations, deductions)."""
    st = np.ascontiguousarray(store, dtype=ITV).copy()
    pr = np.ascontiguousarray(props, dtype=PROP)
    it, de, ent = C.c_uint64(0), C.c_uint64(0), C.c_int(0)
    failed = lib().orc_propagate(st.shape[0], st.ctypes.data, pr.shape[0], pr.ctypes.data,
                                 C.byref(it), C.byref(de), C.byref(ent))
    return st, bool(failed), bool(ent.value), it.value, de.value


def solve(tcn, subproblems_power: int = 0, cutnodes: int = 0, timeout_ms: int = 0,
          stop_after_n_solutions: int = 1, fixed_bound=None, leaf_requires_assignment: int = 0):
    """Full sequential search over a turbo_amd.frontend.TCN-like object.
    leaf_requires_assignment: 0 = barebones' leaf rule (all propagators entailed), 1 = the gpu / cpu paths' (... and every variable assigned).
    Returns (has_solution, best_store, stats_dict)."""
    cfg = OrcConfig(subproblems_power=subproblems_power, has_eps_strategy=int(bool(getattr(tcn, "has_eps_strategy", False))),
                    use_fixed_bound=int(fixed_bound is not None), fixed_bound=int(fixed_bound or 0),
                    stop_after_n_nodes=cutnodes, stop_after_n_solutions=stop_after_n_solutions, timeout_ms=timeout_ms,
                    leaf_requires_assignment=int(leaf_requires_assignment))
    store = np.ascontiguousarray(tcn.store, dtype=ITV)
    props = np.ascontiguousarray(tcn.props, dtype=PROP)
    vo = np.ascontiguousarray(tcn.strat_var_order, dtype=np.int32)
    vl = np.ascontiguousarray(tcn.strat_val_order, dtype=np.int32)
    off = np.ascontiguousarray(tcn.strat_off, dtype=np.int32)
    sv = np.ascontiguousarray(tcn.strat_vars, dtype=np.int32)
    best = np.zeros(store.shape[0], dtype=ITV)
    has = C.c_int32(0)
    stats = OrcStats()
    rc = lib().orc_solve(C.byref(cfg), store.shape[0], store.ctypes.data, props.shape[0], props.ctypes.data,
                         vo.shape[0], vo.ctypes.data, vl.ctypes.data, off.ctypes.data, sv.ctypes.data,
                         int(tcn.obj_var), best.ctypes.data, C.byref(has), C.byref(stats))
    if rc != 0:
        raise RuntimeError(f"orc_solve failed with {rc}")
    return bool(has.value), best, stats.as_dict()


def enumerate_solutions(tcn, capacity: int = 100000, leaf_requires_assignment: int = 0):
    """Every solution leaf of a satisfaction problem, in DFS order (barebones' rule: a box whose unassigned variables are free; the gpu / cpu rule: full assignments)."""
    L = lib()
    L.orc_set_solution_sink.argtypes = [C.c_void_p, C.c_int64]
    L.orc_solution_sink_count.restype = C.c_int64
    n = int(np.asarray(tcn.store).shape[0])
    buf = np.zeros((capacity, max(n, 1)), dtype=ITV)
    L.orc_set_solution_sink(buf.ctypes.data, capacity)
    try:
        _, _, st = solve(tcn, stop_after_n_solutions=0, leaf_requires_assignment=leaf_requires_assignment)
        k = int(L.orc_solution_sink_count())
    finally:
        L.orc_set_solution_sink(None, 0)
    if k > capacity:
        raise RuntimeError(f"{k} solutions exceed the capacity {capacity}")
    return buf[:k, :n], st


def replay_path(tcn, subproblems_power: int, header: dict, decisions: np.ndarray, leaf_requires_assignment: int = 0):
    """Replay a path reported by the HIP engine (capi.Session.debug_path): returns (store under the last node, failed, mismatch) with
    mismatch == -1 when every recorded decision is the one this oracle takes."""
    cfg = OrcConfig(subproblems_power=subproblems_power, has_eps_strategy=int(bool(getattr(tcn, "has_eps_strategy", False))),
                    leaf_requires_assignment=int(leaf_requires_assignment))
    store = np.ascontiguousarray(tcn.store, dtype=ITV)
    props = np.ascontiguousarray(tcn.props, dtype=PROP)
    vo = np.ascontiguousarray(tcn.strat_var_order, dtype=np.int32)
    vl = np.ascontiguousarray(tcn.strat_val_order, dtype=np.int32)
    off = np.ascontiguousarray(tcn.strat_off, dtype=np.int32)
    sv = np.ascontiguousarray(tcn.strat_vars, dtype=np.int32)
    dec = np.ascontiguousarray(decisions)
    assert dec.dtype.itemsize == 28, "orc_path_decision is seven 32-bit words"
    out = np.zeros(store.shape[0], dtype=ITV)
    failed, mismatch = C.c_int32(0), C.c_int32(0)
    L = lib()
    L.orc_replay_path.restype = C.c_int
    L.orc_replay_path.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_int32, C.c_uint64, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = L.orc_replay_path(C.byref(cfg), store.shape[0], store.ctypes.data, props.shape[0], props.ctypes.data,
                           vo.shape[0], vo.ctypes.data, vl.ctypes.data, off.ctypes.data, sv.ctypes.data, int(tcn.obj_var),
                           int(header["subproblem"]), int(header["dive_levels_left"]), int(dec.shape[0]), dec.ctypes.data,
                           int(header["last_objective_ub"]), out.ctypes.data, C.byref(failed), C.byref(mismatch))
    if rc != 0:
        raise RuntimeError(f"orc_replay_path failed with {rc}")
    return out, bool(failed.value), int(mismatch.value)


PATH_DECISION = np.dtype([("var", np.int32), ("child", np.int32), ("lb0", np.int32), ("ub0", np.int32), ("lb1", np.int32), ("ub1", np.int32),
                          ("objective_ub", np.int32)])


class OrcPathHeader(C.Structure):
    _fields_ = [("subproblem", C.c_uint64), ("dive_levels_left", C.c_int32), ("depth", C.c_int32), ("decisions", C.c_int32), ("last_objective_ub", C.c_int32)]


def solve_with_path(tcn, cutnodes: int, subproblems_power: int = 0, capacity: int = 4096, leaf_requires_assignment: int = 0):
    """solve() with a node budget, plus the path the search stood on when it returned (header dict, decisions) and the store under it:
    (has, best, stats, header, decisions, last_store, last_failed)."""
    L = lib()
    L.orc_set_path_sink.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    L.orc_set_node_trace.argtypes = [C.c_void_p, C.c_int64]
    L.orc_set_last_store_sink.argtypes = [C.c_void_p]
    hdr = OrcPathHeader()
    dec = np.zeros(capacity, dtype=PATH_DECISION)
    n = int(np.asarray(tcn.store).shape[0])
    last = np.zeros(max(n, 1), dtype=ITV)
    trace = np.zeros(cutnodes + 1, dtype=np.uint8)
    L.orc_set_path_sink(C.addressof(hdr), dec.ctypes.data, capacity)
    L.orc_set_last_store_sink(last.ctypes.data)
    L.orc_set_node_trace(trace.ctypes.data, trace.shape[0])
    try:
        has, best, st = solve(tcn, subproblems_power=subproblems_power, cutnodes=cutnodes, leaf_requires_assignment=leaf_requires_assignment)
    finally:
        L.orc_set_path_sink(None, None, 0)
        L.orc_set_last_store_sink(None)
        L.orc_set_node_trace(None, 0)
    header = {k: getattr(hdr, k) for k, _ in hdr._fields_}
    return has, best, st, header, dec[:hdr.decisions].copy(), last[:n], bool(trace[st["nodes"] - 1]) if st["nodes"] else False


def solve_traced(tcn, cutnodes: int, subproblems_power: int = 0, leaf_requires_assignment: int = 0):
    """solve() with a node budget, plus the failed flag of every node and the store the search stopped on."""
    L = lib()
    L.orc_set_node_trace.argtypes = [C.c_void_p, C.c_int64]
    L.orc_set_last_store_sink.argtypes = [C.c_void_p]
    n = int(np.asarray(tcn.store).shape[0])
    trace = np.zeros(cutnodes + 1, dtype=np.uint8)
    last = np.zeros(max(n, 1), dtype=ITV)
    L.orc_set_node_trace(trace.ctypes.data, trace.shape[0])
    L.orc_set_last_store_sink(last.ctypes.data)
    try:
        has, best, st = solve(tcn, subproblems_power=subproblems_power, cutnodes=cutnodes, leaf_requires_assignment=leaf_requires_assignment)
    finally:
        L.orc_set_node_trace(None, 0)
        L.orc_set_last_store_sink(None)
    return has, best, st, trace[:st["nodes"]], last[:n]
