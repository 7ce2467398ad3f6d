#!/usr/bin/env python3
"""Synthetic 100k x 500k network (BASELINE.json configs[4], store in global memory): one budgeted search per tb_config given on the command line.
python3 scripts/synth_rate.py "fixpoint=1" "fixpoint=1 debug=0x10100000" ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import capi
from turbo_amd.synth import make_synthetic
tcn = make_synthetic(100_000, 500_000, seed=42)
for spec in sys.argv[1:]:
    kw = dict(a.split("=", 1) for a in spec.split())
    nodes = int(kw.pop("nodes", 8000)); bpc = int(kw.pop("bpc", 0))
    cfg = capi.make_config(stop_after_n_nodes_total=nodes, timeout_ms=120000, **{k: int(v, 0) for k, v in kw.items()})
    cfg.reserved[2] = bpc
    for _ in range(2):
        has, best, st = capi.solve(tcn, cfg)
    secs = st["kernel_ns"] * 1e-9
    print(f"synthetic [{spec}]: {st['num_deductions'] / secs:.4e} props/s  {st['nodes'] / secs:.4e} nodes/s  {st['num_blocks']} x {st['threads_per_block']}  {capi.MEM_KINDS[st['mem_kind']]} {st['shared_bytes']} B", flush=True)
