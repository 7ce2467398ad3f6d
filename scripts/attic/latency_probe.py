# needs the tuning build of the engine: make hip EXTRA_HIPFLAGS=-DTB_TUNING (the knobs are compiled out otherwise)
"""Deterministic single-workgroup latency probe: same tree for every configuration (or_nodes=1, fixed d, cutnodes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import frontend, capi
name = sys.argv[1] if len(sys.argv) > 1 else "example_wordpress7_500.fzn"
cut = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
tcn = frontend.load_fzn(os.path.join(ROOT, "benchmarks", name))
for fp, gm, T in [(2, 0, 1024), (2, 0, 256), (2, 1, 256), (2, 1, 64), (1, 0, 1024), (1, 1, 256)]:
    cfg = capi.make_config(or_nodes=1, subproblems_power=6, stop_after_n_nodes=cut, timeout_ms=120000, fixpoint=fp, threads_per_block=T, only_global_memory=gm)
    cfg.reserved[0] = 0x10000
    has, best, st = capi.solve(tcn, cfg)
    n = st["nodes"]; tot = st["cumulative_time_block_ns"]; t = st["timers_ns"]
    print(f"{name} fp={fp} gm={gm} T={T}: nodes={n} us/node={st['kernel_ns']/n*1e-3:.1f} sweeps/node={st['fixpoint_iterations']/n:.2f} props/node={st['num_deductions']/n:.0f} "
          f"fix%={t[3]/tot:.2f} search%={t[2]/tot:.2f} expand%={t[7]/tot:.2f} own%={t[4]/tot:.2f} wait%={t[5]/tot:.2f} key={(st['fails'], st['solutions'], st['best_bound'])}", flush=True)
