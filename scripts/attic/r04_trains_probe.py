#!/usr/bin/env python3
"""trains15 (and the others) under a tuning of the launch plan: TURBO_HIP_LIB=... python3 scripts/r04_trains_probe.py [instance] [chg_cap ...]
Prints the plan and nodes/s of two 24 M-node steps per change-list capacity (tb_config.reserved[1]; 0 = the engine's choice)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import capi, preprocess
name = sys.argv[1] if len(sys.argv) > 1 else "trains15.fzn"
caps = [int(a) for a in sys.argv[2:]] or [0]
_, tcn, _ = preprocess.load_fzn_simplified(os.path.join(ROOT, "benchmarks", name))
for cap in caps:
    cfg = capi.make_config(fixpoint=2, stop_after_n_nodes_total=24_000_000, timeout_ms=120000)
    cfg.reserved[1] = cap
    s = capi.Session(tcn, cfg)
    pl = s.plan()
    rates = []
    for _ in range(3):
        s.start()
        while not s.poll()[1]:
            pass
        _, _, st = s.finish()
        rates.append(st["nodes"] / (st["kernel_ns"] * 1e-9))
    s.close()
    print(f"{name} lib={os.path.basename(os.environ.get('TURBO_HIP_LIB', 'default'))} chg_cap={cap}: {pl['num_blocks']} x {pl['threads_per_block']} ({pl['num_blocks'] // 256}/CU), {pl['shared_bytes']} B LDS, "
          f"opt {pl['kernel_opt']}: nodes/s {' '.join('%.3e' % r for r in rates)}", flush=True)
