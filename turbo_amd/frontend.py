"""ctypes binding of the host front-end (libturbo_front.so, include/turbo_front.h).

FlatZinc -> ternary constraint network, the stand-in for the reference's
``AbstractDomains::preprocess()`` (include/common_solving.hpp:605-637) in its
``-disable_simplify`` pipeline.  The arrays it returns are exactly what crosses the
C-ABI of the engine (include/turbo_hip.h).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "lib", "libturbo_front.so")

ITV_DTYPE = np.dtype([("lb", np.int32), ("ub", np.int32)])
PROP_DTYPE = np.dtype([("op", np.int32), ("x", np.int32), ("y", np.int32), ("z", np.int32)])

OP_NAMES = ["ADD", "MUL", "TDIV", "TMOD", "MIN", "MAX", "EQ", "LEQ"]
VAR_ORDERS = {"input_order": 0, "first_fail": 1, "anti_first_fail": 2, "smallest": 3, "largest": 4}
VAL_ORDERS = {"min": 0, "max": 1, "split": 2, "reverse_split": 3}

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f"{_LIB_PATH} is missing: build it with `make front` (or `python -c 'import __graft_entry__ as g; g.build()'`)")
        L = C.CDLL(_LIB_PATH)
        L.tf_load_fzn.restype = C.c_void_p
        L.tf_load_fzn.argtypes = [C.c_char_p, C.c_char_p, C.c_int32]
        L.tf_load_fzn_string.restype = C.c_void_p
        L.tf_load_fzn_string.argtypes = [C.c_char_p, C.c_char_p, C.c_int32]
        L.tf_free.argtypes = [C.c_void_p]
        for name in ("tf_load_xcsp3", "tf_load_xcsp3_string"):
            getattr(L, name).restype = C.c_void_p
            getattr(L, name).argtypes = [C.c_char_p, C.c_char_p, C.c_int32]
        L.tf_xcsp3_to_fzn.restype = C.c_int32
        L.tf_xcsp3_to_fzn.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_char_p, C.c_int32]
        for name in ("tf_num_vars", "tf_num_props", "tf_num_strategies", "tf_obj_var", "tf_goal", "tf_goal_var",
                     "tf_trivially_unsat", "tf_parsed_variables", "tf_parsed_constraints"):
            getattr(L, name).restype = C.c_int32
            getattr(L, name).argtypes = [C.c_void_p]
        for name in ("tf_store", "tf_props", "tf_strat_var_order", "tf_strat_val_order", "tf_strat_off", "tf_strat_vars"):
            getattr(L, name).restype = C.c_void_p
            getattr(L, name).argtypes = [C.c_void_p]
        L.tf_push_eps_strategy.restype = C.c_int32
        L.tf_push_eps_strategy.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.tf_objective_of.restype = C.c_int64
        L.tf_objective_of.argtypes = [C.c_void_p, C.c_void_p]
        L.tf_format_solution.restype = C.c_int32
        L.tf_format_solution.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_int32]
        L.tf_simplify.restype = C.c_int32
        L.tf_simplify.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        for name in ("tf_original_num_vars", "tf_original_num_props"):
            getattr(L, name).restype = C.c_int32
            getattr(L, name).argtypes = [C.c_void_p]
        for name in ("tf_original_store", "tf_original_props"):
            getattr(L, name).restype = C.c_void_p
            getattr(L, name).argtypes = [C.c_void_p]
        L.tf_expand_solution.restype = C.c_int32
        L.tf_expand_solution.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.tf_fcn_statistics.restype = C.c_char_p
        L.tf_fcn_statistics.argtypes = [C.c_void_p]
        L.tf_shuffle_strategy.restype = C.c_int32
        L.tf_shuffle_strategy.argtypes = [C.c_void_p, C.c_int32, C.c_uint64]
        L.tf_var_name.restype = C.c_char_p
        L.tf_var_name.argtypes = [C.c_void_p, C.c_int32]
        _lib = L
    return _lib


def _copy(ptr: int, n: int, dtype) -> np.ndarray:
    if n == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


@dataclass
class TCN:
    """The preprocessed problem that crosses the engine's C-ABI (plain arrays)."""
    store: np.ndarray            # ITV_DTYPE[n_vars]
    props: np.ndarray            # PROP_DTYPE[n_props]
    strat_var_order: np.ndarray  # int32[n_strats]
    strat_val_order: np.ndarray  # int32[n_strats]
    strat_off: np.ndarray        # int32[n_strats + 1]
    strat_vars: np.ndarray       # int32[total]
    obj_var: int = -1            # variable to minimise (-1: satisfaction)
    goal: int = 0                # 0 satisfy, 1 minimize, 2 maximize (as written)
    goal_var: int = -1
    trivially_unsat: bool = False
    has_eps_strategy: bool = False
    parsed_variables: int = 0
    parsed_constraints: int = 0
    _model: "Model | None" = field(default=None, repr=False)

    @property
    def n_vars(self) -> int:
        return int(self.store.shape[0])

    @property
    def n_props(self) -> int:
        return int(self.props.shape[0])

    @property
    def n_strats(self) -> int:
        return int(self.strat_var_order.shape[0])

    def objective_of(self, store: np.ndarray) -> int:
        """Objective as the reference prints it (statistics.hpp:378-388)."""
        if self.goal_var < 0:
            return 0
        d = store[self.goal_var]
        return int(d["ub"] if self.goal == 2 else d["lb"])

    def format_solution(self, store: np.ndarray) -> str:
        if self._model is None:
            return ""
        return self._model.format_solution(store)


class Model:
    """Owns a tf_model handle."""

    def __init__(self, handle: int):
        self._h = handle

    @classmethod
    def from_file(cls, path: str) -> "Model":
        err = C.create_string_buffer(1024)
        h = lib().tf_load_fzn(os.fsencode(path), err, len(err))
        if not h:
            raise ValueError(err.value.decode(errors="replace") or "Could not parse input file.")
        return cls(h)

    @classmethod
    def from_string(cls, text: str) -> "Model":
        err = C.create_string_buffer(1024)
        h = lib().tf_load_fzn_string(text.encode(), err, len(err))
        if not h:
            raise ValueError(err.value.decode(errors="replace") or "Could not parse input file.")
        return cls(h)

    @classmethod
    def from_xcsp3_file(cls, path: str) -> "Model":
        err = C.create_string_buffer(1024)
        h = lib().tf_load_xcsp3(os.fsencode(path), err, len(err))
        if not h:
            raise ValueError(err.value.decode(errors="replace") or "Could not parse input file.")
        return cls(h)

    @classmethod
    def from_xcsp3_string(cls, xml: str) -> "Model":
        err = C.create_string_buffer(1024)
        h = lib().tf_load_xcsp3_string(xml.encode(), err, len(err))
        if not h:
            raise ValueError(err.value.decode(errors="replace") or "Could not parse input file.")
        return cls(h)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.tf_free(h)

    def push_eps_strategy(self, var_order: str, val_order: str) -> None:
        if var_order not in VAR_ORDERS:
            raise ValueError(f"Unrecognized option `-eps_var_order {var_order}`")
        if val_order not in VAL_ORDERS:
            raise ValueError(f"Unrecognized option `-eps_value_order {val_order}`")
        lib().tf_push_eps_strategy(self._h, VAR_ORDERS[var_order], VAL_ORDERS[val_order])
        self._eps = True

    def tcn(self) -> TCN:
        L, h = lib(), self._h
        nv, np_, ns = L.tf_num_vars(h), L.tf_num_props(h), L.tf_num_strategies(h)
        off = _copy(L.tf_strat_off(h), ns + 1, np.int32)
        return TCN(
            store=_copy(L.tf_store(h), nv, ITV_DTYPE),
            props=_copy(L.tf_props(h), np_, PROP_DTYPE),
            strat_var_order=_copy(L.tf_strat_var_order(h), ns, np.int32),
            strat_val_order=_copy(L.tf_strat_val_order(h), ns, np.int32),
            strat_off=off,
            strat_vars=_copy(L.tf_strat_vars(h), max(int(off[-1]), 1), np.int32),
            obj_var=L.tf_obj_var(h), goal=L.tf_goal(h), goal_var=L.tf_goal_var(h),
            trivially_unsat=bool(L.tf_trivially_unsat(h)),
            has_eps_strategy=bool(getattr(self, "_eps", False)),
            parsed_variables=L.tf_parsed_variables(h), parsed_constraints=L.tf_parsed_constraints(h),
            _model=self,
        )

    def simplify(self, root_fixpoint: np.ndarray | None = None) -> dict:
        """TCN simplifier (common_solving.hpp:537-585).  `root_fixpoint`: propagated root store of the CURRENT
        network, computed by the caller (GPU engine in production).  Returns the statistics."""
        stats = (C.c_int32 * 9)()
        ptr = None
        if root_fixpoint is not None:
            root_fixpoint = np.ascontiguousarray(root_fixpoint, dtype=ITV_DTYPE)
            assert root_fixpoint.shape[0] == lib().tf_num_vars(self._h)
            ptr = root_fixpoint.ctypes.data
        if lib().tf_simplify(self._h, ptr, stats) != 0:
            raise RuntimeError("tf_simplify failed")
        keys = ["original_vars", "original_props", "simplified_vars", "simplified_props", "merged_variables",
                "cse_merges", "entailed_props", "duplicate_props", "eliminated_variables"]
        return dict(zip(keys, (int(x) for x in stats)))

    def original_network(self):
        """(store, props) of the network as first lowered (what solutions are expanded back to)."""
        L, h = lib(), self._h
        return (_copy(L.tf_original_store(h), L.tf_original_num_vars(h), ITV_DTYPE),
                _copy(L.tf_original_props(h), L.tf_original_num_props(h), PROP_DTYPE))

    def expand_solution(self, store: np.ndarray) -> np.ndarray:
        store = np.ascontiguousarray(store, dtype=ITV_DTYPE)
        out = np.zeros(lib().tf_original_num_vars(self._h), dtype=ITV_DTYPE)
        lib().tf_expand_solution(self._h, store.ctypes.data, out.ctypes.data)
        return out

    def format_solution(self, store: np.ndarray) -> str:
        store = np.ascontiguousarray(store, dtype=ITV_DTYPE)
        n = lib().tf_format_solution(self._h, store.ctypes.data, None, 0)
        buf = C.create_string_buffer(n + 1)
        lib().tf_format_solution(self._h, store.ctypes.data, buf, n + 1)
        return buf.value.decode()

    def fcn_statistics(self) -> dict:
        """analyze_cn statistics of the model as parsed (key -> text)."""
        out = {}
        for line in lib().tf_fcn_statistics(self._h).decode().splitlines():
            k, _, v = line.partition("=")
            out[k] = v.strip('"')
        return out

    def shuffle_strategy(self, strategy: int, seed: int) -> None:
        """`-eps_var_order random`: INPUT_ORDER over the strategy's variables shuffled with mt19937(seed)."""
        if lib().tf_shuffle_strategy(self._h, strategy, seed) != 0:
            raise ValueError("no such strategy")

    def var_name(self, v: int) -> str:
        return lib().tf_var_name(self._h, v).decode()


def xcsp3_to_fzn(xml: str) -> str:
    """The FlatZinc text an XCSP3 instance is rewritten to (debugging aid)."""
    err = C.create_string_buffer(1024)
    n = lib().tf_xcsp3_to_fzn(xml.encode(), None, 0, err, len(err))
    if n < 0:
        raise ValueError(err.value.decode(errors="replace"))
    buf = C.create_string_buffer(n + 1)
    lib().tf_xcsp3_to_fzn(xml.encode(), buf, n + 1, err, len(err))
    return buf.value.decode()


def load_fzn(path: str, eps_var_order: str = "default", eps_value_order: str = "default") -> TCN:
    """Parse + lower a FlatZinc file; optional EPS strategy like `-eps_var_order/-eps_value_order`."""
    m = Model.from_file(path)
    if (eps_var_order == "default") != (eps_value_order == "default"):
        raise ValueError("-eps_var_order and -eps_value_order must be specified together.")
    if eps_var_order != "default":
        m.push_eps_strategy(eps_var_order, eps_value_order)
    return m.tcn()
