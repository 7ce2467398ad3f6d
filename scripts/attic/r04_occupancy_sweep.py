#!/usr/bin/env python3
"""nodes/s of one instance against the workgroups per CU (tb_config.reserved[2] caps them) for 256- and 128-thread workgroups:
python3 scripts/r04_occupancy_sweep.py [instance]  -- the curve an LDS-halving layout tier would move along."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import capi, preprocess
name = sys.argv[1] if len(sys.argv) > 1 else "trains15.fzn"
_, tcn, _ = preprocess.load_fzn_simplified(os.path.join(ROOT, "benchmarks", name))
for T, caps in ((256, (3, 4, 5, 6, 7)), (128, (4, 5, 6, 7, 8, 10, 12, 14))):
    for bpc in caps:
        cfg = capi.make_config(fixpoint=2, stop_after_n_nodes_total=16_000_000, timeout_ms=120000, threads_per_block=T)
        cfg.reserved[2] = bpc
        try:
            s = capi.Session(tcn, cfg)
        except Exception as e:
            print(T, bpc, "ERR", e); continue
        pl = s.plan()
        rates = []
        for _ in range(2):
            s.start()
            while not s.poll()[1]:
                pass
            _, _, st = s.finish()
            rates.append(st["nodes"] / (st["kernel_ns"] * 1e-9))
        s.close()
        print(f"{name} T={T} cap={bpc}: {pl['num_blocks']} x {pl['threads_per_block']} ({pl['num_blocks'] // 256}/CU = {pl['num_blocks'] // 256 * pl['threads_per_block'] // 64} waves/CU), {pl['shared_bytes']} B LDS, "
              f"mem {pl['mem_kind']}: nodes/s {' '.join('%.3e' % r for r in rates)}", flush=True)
