"""Event mode: nodes/s against the workgroup width (how much intra-node parallelism the worklist exposes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import capi, preprocess
names = sys.argv[1:] or ["example_wordpress7_500.fzn", "trains15.fzn", "accap_a3.fzn"]
for name in names:
    _, tcn, _ = preprocess.load_fzn_simplified(os.path.join(ROOT, "benchmarks", name))
    for T in (64, 128, 256, 512, 1024):
        cfg = capi.make_config(fixpoint=2, timeout_ms=20000, threads_per_block=T, stop_after_n_nodes=4000)
        for _ in range(2):
            has, sol, st = capi.solve(tcn, cfg)
        secs = st["kernel_ns"] * 1e-9
        n = st["nodes"]
        print(f"{name:28s} T={T:5d}: blocks={st['num_blocks']}x{st['threads_per_block']} mem={capi.MEM_KINDS[st['mem_kind']]} lds={st['shared_bytes']} "
              f"{n/secs:.3e} nodes/s props/node={st['num_deductions']/max(1,n):.0f} us/node/block={secs*1e6*st['num_blocks']/n:.1f}", flush=True)
