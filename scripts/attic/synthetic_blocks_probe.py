"""Synthetic 100k x 500k (store in global memory): propagations/s against the number of workgroups (working set vs Infinity Cache)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import capi
from turbo_amd.synth import make_synthetic
tcn = make_synthetic(100_000, 500_000, seed=42)
for fp in (1, 2):
    for blocks, T in ((128, 1024), (256, 1024), (384, 1024), (512, 1024), (256, 512), (512, 512), (1024, 256), (0, 0)):
        cfg = capi.make_config(fixpoint=fp, timeout_ms=60000, stop_after_n_nodes=40, or_nodes=blocks, threads_per_block=T)
        for _ in range(2):
            has, sol, st = capi.solve(tcn, cfg)
        secs = st["kernel_ns"] * 1e-9
        print(f"fp={fp} or={blocks:5d} T={T:5d}: blocks={st['num_blocks']}x{st['threads_per_block']} mem={capi.MEM_KINDS[st['mem_kind']]} "
              f"{st['num_deductions']/secs:.3e} props/s {st['nodes']/secs:.3e} nodes/s  alg {st['num_deductions']*40/secs/1e9:.0f} GB/s", flush=True)
