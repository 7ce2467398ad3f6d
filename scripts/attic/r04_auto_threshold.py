#!/usr/bin/env python3
"""Which fixpoint `-fp auto` should pick on small networks: kernel time of a full solve (or a 2 M-node budget) with WAC1 sweeps and with the event fixpoint,
on the reference's regression instances and accap_a3 (simplified networks, the CLI's default pipeline)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import capi, preprocess
names = ["test_data/sudoku_opt4.fzn", "test_data/pat2.fzn", "test_data/pat7.fzn", "test_data/pennies5.fzn", "test_data/triangular9.fzn", "test_data/sudoku_opt3.fzn",
         "test_data/bug4.fzn", "test_data/reified_in.fzn", "accap_a3.fzn"]
for name in names:
    _, tcn, _ = preprocess.load_fzn_simplified(os.path.join(ROOT, "benchmarks", name))
    row = []
    for fp in (1, 2):
        cfg = capi.make_config(fixpoint=fp, stop_after_n_nodes_total=2_000_000, timeout_ms=60000)
        capi.solve(tcn, cfg)
        _, _, st = capi.solve(tcn, cfg)
        row.append((st["nodes"], st["kernel_ns"] * 1e-6, st["nodes"] / max(1e-9, st["kernel_ns"] * 1e-9), st["num_blocks"], st["threads_per_block"]))
    print(f"{name:32s} {tcn.n_vars:6d} x {tcn.n_props:6d}: wac1 {row[0][0]:8d} nodes {row[0][1]:8.2f} ms {row[0][2]:.3e}/s ({row[0][3]} x {row[0][4]}) | event {row[1][0]:8d} nodes {row[1][1]:8.2f} ms {row[1][2]:.3e}/s ({row[1][3]} x {row[1][4]})", flush=True)
