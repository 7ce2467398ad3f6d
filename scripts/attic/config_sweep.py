"""Try launch configurations (memory kind x workgroup size x fixpoint) on one instance, same EPS decomposition."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import frontend, capi
name = sys.argv[1] if len(sys.argv) > 1 else "example_wordpress7_500.fzn"
power = int(sys.argv[2]) if len(sys.argv) > 2 else 18
fps = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [2, 1]
tcn = frontend.load_fzn(os.path.join(ROOT, "benchmarks", name))
for fp in fps:
    for gm, T, cap, bpc in [(0, 1024, 0, 0), (1, 256, 0, 0), (1, 256, 256, 0), (1, 128, 256, 16), (1, 64, 256, 16), (1, 64, 128, 32), (1, 128, 128, 16)]:
        if True:
            try:
                cfg = capi.make_config(timeout_ms=2500, fixpoint=fp, only_global_memory=gm, threads_per_block=T, subproblems_power=power)
                cfg.reserved[1] = cap; cfg.reserved[2] = bpc
                has, best, st = capi.solve(tcn, cfg)
            except Exception as e:
                print(fp, gm, T, "ERR", e); continue
            secs = st["kernel_ns"] * 1e-9
            print(f"{name} d={power} fp={fp} globalmem={gm} T={T} cap={cap} bpc={bpc}: blocks={st['num_blocks']} mem={capi.MEM_KINDS[st['mem_kind']]} nodes/s={st['nodes']/secs:.3e} props/s={st['num_deductions']/secs:.3e} sweeps/node={st['fixpoint_iterations']/max(1,st['nodes']):.1f} props/node={st['num_deductions']/max(1,st['nodes']):.0f}", flush=True)
