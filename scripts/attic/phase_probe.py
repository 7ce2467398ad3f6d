# needs the tuning build of the engine: make hip EXTRA_HIPFLAGS=-DTB_TUNING (the knobs are compiled out otherwise)
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import frontend, capi
name = sys.argv[1] if len(sys.argv) > 1 else "example_wordpress7_500.fzn"
tcn = frontend.load_fzn(os.path.join(ROOT, "benchmarks", name))
for fp in (2,):
    cfg = capi.make_config(fixpoint=fp, timeout_ms=3000)
    cfg.reserved[0] = 0x10000
    has, best, st = capi.solve(tcn, cfg)
    tot = st["cumulative_time_block_ns"]
    t = st["timers_ns"]
    n = st["nodes"]
    print(f"{name} fp={fp}: nodes={n} ({n/(st['kernel_ns']*1e-9):.3e}/s) sweeps/node={st['fixpoint_iterations']/n:.1f} props/node={st['num_deductions']/n:.0f} "
          f"fix%={t[3]/tot:.2f} search%={t[2]/tot:.2f} own_work%={t[4]/tot:.2f} barrier_wait%={t[5]/tot:.2f} dive%={t[8]/tot:.2f} snap_push%={t[1]/tot:.2f} split%={t[6]/tot:.2f} us/node/block={tot/n*1e-3:.1f}")
