"""A/B probe of event-mode options on fixed work (cutnodes): raw vs simplified network, debug bits on/off.

usage: python scripts/event_ab.py [--bits 0x20000] [--cut N] [instances...]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import capi, frontend, preprocess

ap = argparse.ArgumentParser()
ap.add_argument("--bits", type=lambda s: int(s, 0), default=0x20000)
ap.add_argument("--cut", type=int, default=6000)
ap.add_argument("--fixpoint", type=int, default=2)
ap.add_argument("instances", nargs="*", default=["example_wordpress7_500.fzn", "accap_a3.fzn", "trains15.fzn"])
a = ap.parse_args()

for name in a.instances:
    path = os.path.join(ROOT, "benchmarks", name)
    raw = frontend.load_fzn(path)
    _, simp, _ = preprocess.load_fzn_simplified(path)
    for label, tcn in (("raw", raw), ("simplified", simp)):
        for bits in (0, a.bits):
            cfg = capi.make_config(fixpoint=a.fixpoint, timeout_ms=20000, stop_after_n_nodes=a.cut)
            cfg.reserved[0] = bits
            best = None
            for rep in range(2):  # second run: warm code objects / clocks
                has, sol, st = capi.solve(tcn, cfg)
            secs = st["kernel_ns"] * 1e-9
            n = st["nodes"]
            print(f"{name:28s} {label:10s} V={tcn.n_vars:6d} P={tcn.n_props:6d} bits={bits:#x}: blocks={st['num_blocks']}x{st['threads_per_block']} "
                  f"mem={capi.MEM_KINDS[st['mem_kind']]} nodes={n} {n/secs:.3e} nodes/s props/node={st['num_deductions']/max(1,n):.0f} "
                  f"{st['num_deductions']/secs:.3e} props/s t={secs*1e3:.0f} ms", flush=True)
