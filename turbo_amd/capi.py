"""ctypes binding of the engine's C-ABI (libturbo_hip.so, include/turbo_hip.h).

This is the Python mirror of what a host program binds; it contains no solving logic and has
NO CPU fallback: if the HIP library or a GPU is missing every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .frontend import ITV_DTYPE, PROP_DTYPE

_HERE = os.path.dirname(os.path.abspath(__file__))
# TURBO_HIP_LIB: another build of the same engine (e.g. `make tuning`: -DTB_TUNING, the in-kernel profiling knobs compiled in)
LIB_PATH = os.environ.get("TURBO_HIP_LIB") or os.path.join(_HERE, "lib", "libturbo_hip.so")

TB_PINF = 2**31 - 1
TB_NINF = -(2**31)
MEM_KINDS = ["global", "store_shared", "tcn_shared"]
TIMERS = ["OVERALL", "PREPROCESSING", "SEARCH", "FIXPOINT", "TRANSFER_CPU2GPU", "TRANSFER_GPU2CPU",
          "SELECT_FP_FUNCTIONS", "WAIT_CPU", "DIVE", "LATEST_BEST_OBJ_FOUND", "FIRST_BLOCK_IDLE"]
PROF = ["SEEDING", "ROUNDS", "SNAPSHOT_PUSH", "VARIABLE_SELECTION"]  # tb_stats.prof_ns (tuning build, knob 0x10000)

EXPORTS = ["tb_version", "tb_last_error", "tb_device_count", "tb_get_device_info", "tb_eps_local_count", "tb_eps_global_index",
           "tb_propagate", "tb_solve",
           "tb_session_create", "tb_session_start", "tb_session_poll", "tb_session_push_bound",
           "tb_session_stop", "tb_session_next_solution", "tb_session_finish", "tb_session_destroy",
           "tb_session_export_peer", "tb_session_import_peer", "tb_session_link_peer", "tb_session_arm",
           "tb_session_progress", "tb_session_debug_last_store", "tb_session_plan", "tb_session_unlink_peers", "tb_session_debug_path"]


class TbConfig(C.Structure):
    _fields_ = [("timeout_ms", C.c_uint64), ("or_nodes", C.c_uint64), ("subproblems_factor", C.c_uint64),
                ("stop_after_n_nodes", C.c_uint64), ("stop_after_n_solutions", C.c_uint64),
                ("wac1_threshold", C.c_uint64), ("stop_after_n_nodes_total", C.c_uint64),
                ("subproblems_power", C.c_int32), ("fixpoint", C.c_int32), ("only_global_memory", C.c_int32),
                ("verbose", C.c_int32), ("has_eps_strategy", C.c_int32), ("threads_per_block", C.c_int32),
                ("device", C.c_int32), ("rank", C.c_int32), ("world_size", C.c_int32),
                ("use_fixed_bound", C.c_int32), ("fixed_bound", C.c_int32), ("deterministic", C.c_int32),
                ("snapshot_levels", C.c_int32), ("stream_solutions", C.c_int32), ("entailed_prop_removal", C.c_int32),
                ("eps_chunk_log2", C.c_int32), ("decision_stack_depth", C.c_int32), ("poll_period_us", C.c_int32),
                ("reserved", C.c_int32 * 3), ("leaf_requires_assignment", C.c_int32)]


class TbStats(C.Structure):
    _fields_ = [("nodes", C.c_uint64), ("fails", C.c_uint64), ("solutions", C.c_uint64),
                ("fixpoint_iterations", C.c_uint64), ("num_deductions", C.c_uint64),
                ("eps_num_subproblems", C.c_uint64), ("eps_solved_subproblems", C.c_uint64),
                ("eps_skipped_subproblems", C.c_uint64), ("num_blocks_done", C.c_uint64),
                ("timers_ns", C.c_int64 * 11), ("cumulative_time_block_ns", C.c_int64), ("kernel_ns", C.c_int64),
                ("store_writes", C.c_uint64),
                ("depth_max", C.c_int32), ("num_blocks", C.c_int32), ("threads_per_block", C.c_int32),
                ("exhaustive", C.c_int32), ("mem_kind", C.c_int32), ("shared_bytes", C.c_int32),
                ("subproblems_power", C.c_int32), ("best_bound", C.c_int32), ("best_subproblem", C.c_int32),
                ("interrupted", C.c_int32), ("reserved", C.c_int32 * 2),
                ("eps_local_subproblems", C.c_uint64), ("eps_stolen_subproblems", C.c_uint64), ("wait_time_ns", C.c_int64),
                ("min_block_ns", C.c_int64), ("max_block_ns", C.c_int64), ("active_lane_evaluations", C.c_uint64),
                ("prof_ns", C.c_int64 * 4)]

    def as_dict(self) -> dict:
        d = {}
        for k, _ in self._fields_:
            if k == "reserved":
                d["why_not_exhaustive"] = int(self.reserved[0])
                d["debug_slice"] = int(self.reserved[1])
                continue
            v = getattr(self, k)
            d[k] = list(v) if hasattr(v, "__len__") else v
        return d


class TbPlan(C.Structure):
    _fields_ = [("num_blocks", C.c_int32), ("threads_per_block", C.c_int32), ("mem_kind", C.c_int32), ("shared_bytes", C.c_int32),
                ("subproblems_power", C.c_int32), ("eps_chunk_log2", C.c_int32), ("snapshot_levels", C.c_int32),
                ("decision_stack_depth", C.c_int32), ("eps_local_subproblems", C.c_uint64),
                ("kernel_event", C.c_int32), ("kernel_opt", C.c_int32)]


class TbDebugPath(C.Structure):
    _fields_ = [("subproblem", C.c_uint64), ("dive_levels_left", C.c_int32), ("depth", C.c_int32), ("decisions", C.c_int32),
                ("last_objective_ub", C.c_int32), ("last_node_failed", C.c_int32), ("had_work", C.c_int32), ("nodes", C.c_int32), ("reserved", C.c_int32)]


DEBUG_DECISION_DTYPE = np.dtype([("var", np.int32), ("child", np.int32), ("lb0", np.int32), ("ub0", np.int32), ("lb1", np.int32), ("ub1", np.int32),
                                 ("objective_ub", np.int32)])


class TbDeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 256), ("compute_units", C.c_int32), ("lds_bytes_per_cu", C.c_int32),
                ("wavefront_size", C.c_int32), ("clock_khz", C.c_int32), ("total_global_mem", C.c_int64),
                ("is_gfx950", C.c_int32), ("xcc_count", C.c_int32)]


class TurboHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libturbo_hip error {code}: {msg}")
        self.code = code


_lib = None


def lib() -> C.CDLL:
    """Load libturbo_hip.so (built in-tree by `make hip`).  Fails loudly when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: the HIP engine is not built (run `make hip`); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.tb_version.restype = C.c_char_p
        L.tb_last_error.restype = C.c_char_p
        L.tb_device_count.restype = C.c_int
        L.tb_get_device_info.restype = C.c_int
        L.tb_get_device_info.argtypes = [C.c_int, C.POINTER(TbDeviceInfo)]
        L.tb_eps_local_count.restype = C.c_int
        L.tb_eps_local_count.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint64)]
        L.tb_eps_global_index.restype = C.c_int
        L.tb_eps_global_index.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.POINTER(C.c_uint64)]
        L.tb_propagate.restype = C.c_int
        L.tb_propagate.argtypes = [C.POINTER(TbConfig), C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
        net = [C.POINTER(TbConfig), C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
               C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        L.tb_solve.restype = C.c_int
        L.tb_solve.argtypes = net + [C.POINTER(C.c_int32), C.c_void_p, C.POINTER(C.c_int32), C.POINTER(TbStats)]
        L.tb_session_create.restype = C.c_int
        L.tb_session_create.argtypes = net + [C.POINTER(C.c_void_p)]
        L.tb_session_plan.restype = C.c_int
        L.tb_session_plan.argtypes = [C.c_void_p, C.POINTER(TbPlan)]
        L.tb_session_export_peer.restype = C.c_int
        L.tb_session_export_peer.argtypes = [C.c_void_p, C.c_void_p]
        L.tb_session_import_peer.restype = C.c_int
        L.tb_session_import_peer.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.tb_session_link_peer.restype = C.c_int
        L.tb_session_link_peer.argtypes = [C.c_void_p, C.c_void_p]
        L.tb_session_progress.restype = C.c_int
        L.tb_session_progress.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.tb_session_debug_last_store.restype = C.c_int
        L.tb_session_debug_last_store.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        # (TURBO_HIP_LIB may name an older build of the engine for a same-box A/B: entry points added since are bound when present)
        if hasattr(L, "tb_session_debug_path"):
            L.tb_session_debug_path.restype = C.c_int
            L.tb_session_debug_path.argtypes = [C.c_void_p, C.c_int32, C.POINTER(TbDebugPath), C.c_int32, C.c_void_p]
        for name in ("tb_session_start", "tb_session_stop", "tb_session_arm", "tb_session_unlink_peers"):
            if name == "tb_session_unlink_peers" and not hasattr(L, name):
                continue
            getattr(L, name).restype = C.c_int
            getattr(L, name).argtypes = [C.c_void_p]
        L.tb_session_poll.restype = C.c_int
        L.tb_session_poll.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.tb_session_push_bound.restype = C.c_int
        L.tb_session_push_bound.argtypes = [C.c_void_p, C.c_int32]
        L.tb_session_next_solution.restype = C.c_int
        L.tb_session_next_solution.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.tb_session_finish.restype = C.c_int
        L.tb_session_finish.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(TbStats)]
        L.tb_session_destroy.restype = None
        L.tb_session_destroy.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise TurboHipError(rc, lib().tb_last_error().decode(errors="replace"))


def make_config(**kw) -> TbConfig:
    """Configuration with the reference's defaults (include/config.hpp:61-105)."""
    cfg = TbConfig()
    cfg.subproblems_power = -1
    cfg.subproblems_factor = 300
    cfg.stop_after_n_solutions = 1
    cfg.fixpoint = 1  # WAC1 is the GPU default (config.hpp:91-97)
    cfg.world_size = 1
    if "debug" in kw:  # tuning / test knobs of tb_config.reserved[0] (e.g. 0x100000: force the COMPACT store layout)
        cfg.reserved[0] = int(kw.pop("debug"))
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise TypeError(f"unknown tb_config field {k}")
        setattr(cfg, k, v)
    return cfg


def device_info(device: int = 0) -> dict:
    info = TbDeviceInfo()
    check(lib().tb_get_device_info(device, C.byref(info)))
    return {k: (getattr(info, k).decode() if k == "name" else getattr(info, k)) for k, _ in info._fields_}


def eps_local_count(subproblems_power: int, chunk_log2: int, rank: int, world_size: int) -> int:
    """Size of `rank`'s block-cyclic share of the 2^d subproblems."""
    n = C.c_uint64(0)
    check(lib().tb_eps_local_count(subproblems_power, chunk_log2, rank, world_size, C.byref(n)))
    return n.value


def eps_global_index(subproblems_power: int, chunk_log2: int, rank: int, world_size: int, j: int) -> int:
    """Global index of the j-th subproblem of `rank`."""
    g = C.c_uint64(0)
    check(lib().tb_eps_global_index(subproblems_power, chunk_log2, rank, world_size, j, C.byref(g)))
    return g.value


def _net_args(tcn):
    store = np.ascontiguousarray(tcn.store, dtype=ITV_DTYPE)
    props = np.ascontiguousarray(tcn.props, dtype=PROP_DTYPE)
    vo = np.ascontiguousarray(tcn.strat_var_order, dtype=np.int32)
    vl = np.ascontiguousarray(tcn.strat_val_order, dtype=np.int32)
    off = np.ascontiguousarray(tcn.strat_off, dtype=np.int32)
    sv = np.ascontiguousarray(tcn.strat_vars, dtype=np.int32)
    keep = (store, props, vo, vl, off, sv)
    args = [store.shape[0], store.ctypes.data, props.shape[0], props.ctypes.data,
            vo.shape[0], vo.ctypes.data, vl.ctypes.data, off.ctypes.data, sv.ctypes.data, int(tcn.obj_var)]
    return keep, args


def propagate(props: np.ndarray, stores: np.ndarray, cfg: TbConfig | None = None):
    """Batch node propagation on the GPU.  stores: ITV_DTYPE[n_stores, n_vars].
    Returns (stores_out, failed[int32], all_entailed[int32], iterations[u64], deductions[u64], kernel_ns)."""
    cfg = cfg or make_config()
    props = np.ascontiguousarray(props, dtype=PROP_DTYPE)
    stores = np.ascontiguousarray(stores, dtype=ITV_DTYPE).copy()
    if stores.ndim == 1:
        stores = stores[None, :]
    n_stores, n_vars = stores.shape
    failed = np.zeros(n_stores, dtype=np.int32)
    ent = np.zeros(n_stores, dtype=np.int32)
    iters = np.zeros(n_stores, dtype=np.uint64)
    ded = np.zeros(n_stores, dtype=np.uint64)
    ns = C.c_int64(0)
    check(lib().tb_propagate(C.byref(cfg), n_vars, props.shape[0], props.ctypes.data, n_stores, stores.ctypes.data,
                             failed.ctypes.data, ent.ctypes.data, iters.ctypes.data, ded.ctypes.data, C.byref(ns)))
    return stores, failed, ent, iters, ded, ns.value


def solve(tcn, cfg: TbConfig | None = None, stop_flag: C.c_int32 | None = None):
    """Blocking dive-and-solve.  Returns (has_solution, best_store, stats_dict)."""
    cfg = cfg or make_config()
    cfg.has_eps_strategy = int(bool(getattr(tcn, "has_eps_strategy", False)))
    keep, args = _net_args(tcn)
    best = np.zeros(max(tcn.n_vars, 1), dtype=ITV_DTYPE)
    has = C.c_int32(0)
    stats = TbStats()
    flag = stop_flag if stop_flag is not None else C.c_int32(0)
    check(lib().tb_solve(C.byref(cfg), *args, C.byref(flag), best.ctypes.data, C.byref(has), C.byref(stats)))
    del keep
    return bool(has.value), best[:tcn.n_vars], stats.as_dict()


class Session:
    """Asynchronous solve on one device (what the multi-GPU driver uses)."""

    def __init__(self, tcn, cfg: TbConfig):
        cfg.has_eps_strategy = int(bool(getattr(tcn, "has_eps_strategy", False)))
        self._keep, args = _net_args(tcn)
        self._n_vars = tcn.n_vars
        self._h = C.c_void_p()
        check(lib().tb_session_create(C.byref(cfg), *args, C.byref(self._h)))

    def start(self) -> None:
        check(lib().tb_session_start(self._h))

    def plan(self) -> dict:
        """Grid, memory kind and 2^d chosen by tb_session_create."""
        pl = TbPlan()
        check(lib().tb_session_plan(self._h, C.byref(pl)))
        return {k: getattr(pl, k) for k, _ in pl._fields_}

    def arm(self) -> None:
        """Reset the device-side state of a search (linked sessions: every rank arms, all synchronise, every rank starts)."""
        check(lib().tb_session_arm(self._h))

    def export_peer(self) -> bytes:
        """64-byte IPC handle of this session's cell (for tb_session_import_peer in another process)."""
        buf = C.create_string_buffer(64)
        check(lib().tb_session_export_peer(self._h, buf))
        return buf.raw

    def import_peer(self, peer_rank: int, handle: bytes) -> None:
        buf = C.create_string_buffer(bytes(handle), 64)
        check(lib().tb_session_import_peer(self._h, int(peer_rank), buf))

    def link_peer(self, other: "Session") -> None:
        """Same process: let this session's kernel reach `other`'s cell (call it in both directions)."""
        check(lib().tb_session_link_peer(self._h, other._h))

    def unlink_peers(self) -> None:
        """Forget every imported / linked cell (a group that cannot be linked completely uses the host relay as a whole)."""
        check(lib().tb_session_unlink_peers(self._h))

    def progress(self, counters: bool = False):
        """Remaining subproblems of this GPU as of the kernel's last poll (and, with counters, stolen in / out so far)."""
        rem, si, so = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        check(lib().tb_session_progress(self._h, C.byref(rem), C.byref(si) if counters else None, C.byref(so) if counters else None))
        return (rem.value, si.value, so.value) if counters else rem.value

    def debug_last_store(self, workgroup: int = 0):
        out = np.zeros(max(self._n_vars, 1), dtype=ITV_DTYPE)
        check(lib().tb_session_debug_last_store(self._h, workgroup, out.ctypes.data))
        return out[:self._n_vars]

    def debug_path(self, workgroup: int = 0, capacity: int = 4096):
        """Test aid: (header dict, decisions[DEBUG_DECISION_DTYPE]) of the path workgroup `workgroup` stood on when it left the kernel."""
        hdr = TbDebugPath()
        dec = np.zeros(max(capacity, 1), dtype=DEBUG_DECISION_DTYPE)
        check(lib().tb_session_debug_path(self._h, workgroup, C.byref(hdr), capacity, dec.ctypes.data))
        return {k: getattr(hdr, k) for k, _ in hdr._fields_}, dec[:hdr.decisions]

    def poll(self):
        best, done = C.c_int32(0), C.c_int32(0)
        check(lib().tb_session_poll(self._h, C.byref(best), C.byref(done)))
        return best.value, bool(done.value)

    def push_bound(self, bound: int) -> None:
        check(lib().tb_session_push_bound(self._h, int(bound)))

    def stop(self) -> None:
        check(lib().tb_session_stop(self._h))

    def next_solution(self):
        """Streaming (cfg.stream_solutions=1): (store, objective) of the next solution handed over, or None."""
        out = np.zeros(max(self._n_vars, 1), dtype=ITV_DTYPE)
        obj, has = C.c_int32(0), C.c_int32(0)
        check(lib().tb_session_next_solution(self._h, out.ctypes.data, C.byref(obj), C.byref(has)))
        return (out[:self._n_vars], obj.value) if has.value else None

    def finish(self):
        best = np.zeros(max(self._n_vars, 1), dtype=ITV_DTYPE)
        has = C.c_int32(0)
        stats = TbStats()
        check(lib().tb_session_finish(self._h, best.ctypes.data, C.byref(has), C.byref(stats)))
        return bool(has.value), best[:self._n_vars], stats.as_dict()

    def close(self) -> None:
        if self._h:
            lib().tb_session_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
