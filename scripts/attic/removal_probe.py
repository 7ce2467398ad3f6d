"""Sweeping fixpoints with and without entailed-slice removal (tb_config.entailed_prop_removal) on fixed work."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import capi, preprocess
for name in sys.argv[1:] or ["example_wordpress7_500.fzn", "accap_a3.fzn", "trains15.fzn"]:
    _, tcn, _ = preprocess.load_fzn_simplified(os.path.join(ROOT, "benchmarks", name))
    for fp in (1, 0):
        for rm in (0, 1):
            cfg = capi.make_config(fixpoint=fp, timeout_ms=60000, stop_after_n_nodes=3000, entailed_prop_removal=rm)
            for _ in range(2):
                has, sol, st = capi.solve(tcn, cfg)
            secs, n = st["kernel_ns"] * 1e-9, st["nodes"]
            print(f"{name:28s} fp={'wac1' if fp else 'ac1 '} removal={rm}: blocks={st['num_blocks']}x{st['threads_per_block']} {n/secs:.3e} nodes/s "
                  f"props/node={st['num_deductions']/max(1,n):.0f} {st['num_deductions']/secs:.3e} props/s t={secs*1e3:.0f} ms", flush=True)
