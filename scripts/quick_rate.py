#!/usr/bin/env python3
"""One budgeted search with arbitrary tb_config fields, for quick A/B runs on the GPU box:
python3 scripts/quick_rate.py trains15 nodes=8000000 fixpoint=2 only_global_memory=1 [debug=0x100000] [raw=1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import capi, frontend, preprocess
wl = sys.argv[1]
kw = dict(a.split("=", 1) for a in sys.argv[2:])
nodes = int(kw.pop("nodes", 8_000_000)); raw = int(kw.pop("raw", 0)); reps = int(kw.pop("reps", 2)); bpc = int(kw.pop("bpc", 0))
fzn = {"wordpress7_500": "example_wordpress7_500.fzn", "accap_a3": "accap_a3.fzn", "trains15": "trains15.fzn"}[wl]
path = os.path.join(ROOT, "benchmarks", fzn)
tcn = frontend.load_fzn(path) if raw else preprocess.load_fzn_simplified(path)[1]
cfg = capi.make_config(stop_after_n_nodes_total=nodes, timeout_ms=120000, **{k: int(v, 0) for k, v in kw.items()})
cfg.reserved[2] = bpc  # cap on workgroups per CU (tuning knob)
for _ in range(reps):
    has, best, st = capi.solve(tcn, cfg)
print(f"{wl} {kw}: {st['nodes'] / (st['kernel_ns'] * 1e-9):.4e} nodes/s  {st['num_deductions'] / (st['kernel_ns'] * 1e-9):.4e} props/s  {st['num_blocks']} x {st['threads_per_block']}  {capi.MEM_KINDS[st['mem_kind']]} {st['shared_bytes']} B  2^{st['subproblems_power']}", flush=True)
