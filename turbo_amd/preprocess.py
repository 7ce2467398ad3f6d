"""Preprocessing driver: FlatZinc -> TCN -> (root propagation on the GPU -> simplifier) x rounds.

Mirrors the loop of AbstractDomains::preprocess_tcn (include/common_solving.hpp:537-585).  The root propagation
is done by the engine itself (`tb_propagate`), so the propagators stay single-sourced; the simplifier
(libturbo_front, simplify.cpp) only manipulates the network.
"""
from __future__ import annotations

from . import capi, frontend


def load_fzn_simplified(path: str, rounds: int = 16, eps_var_order: str = "default", eps_value_order: str = "default",
                        propagate=None, device: int = 0):
    """Returns (model, tcn, stats).  `propagate(store, props) -> (store, failed)` defaults to the GPU engine."""
    m = frontend.Model.from_file(path)
    if propagate is None:
        def propagate(store, props):
            out, failed, _, _, _, _ = capi.propagate(props, store[None, :], capi.make_config(fixpoint=1, device=device))
            return out[0], bool(failed[0])
    stats = []
    for _ in range(rounds):
        tcn = m.tcn()
        if tcn.trivially_unsat:
            break
        root, failed = propagate(tcn.store, tcn.props)
        if failed:
            # an inconsistent root: hand the failing store over, the simplifier records the UNSAT status
            st = m.simplify(root)
            stats.append(st)
            break
        st = m.simplify(root)
        stats.append(st)
        if not any(st[k] for k in ("merged_variables", "cse_merges", "entailed_props", "duplicate_props", "eliminated_variables")):
            break
    if eps_var_order != "default":
        m.push_eps_strategy(eps_var_order, eps_value_order)
    return m, m.tcn(), stats
