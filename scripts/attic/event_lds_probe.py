"""Event mode: store in LDS (few wide workgroups per CU) against store in global memory (many narrow ones)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import capi, frontend, preprocess
names = sys.argv[1:] or ["example_wordpress7_500.fzn", "trains15.fzn"]
for name in names:
    path = os.path.join(ROOT, "benchmarks", name)
    _, tcn, _ = preprocess.load_fzn_simplified(path)
    for bits, T in ((0, 0), (0x40000, 256), (0x40000, 512), (0x40000, 1024)):
        cfg = capi.make_config(fixpoint=2, timeout_ms=3000, threads_per_block=T)
        cfg.reserved[0] = bits
        try:
            has, sol, st = capi.solve(tcn, cfg)
        except Exception as e:
            print(name, hex(bits), T, "error", e); continue
        secs = st["kernel_ns"] * 1e-9
        n = st["nodes"]
        print(f"{name:28s} V={tcn.n_vars} P={tcn.n_props} bits={bits:#x} T={T}: blocks={st['num_blocks']}x{st['threads_per_block']} mem={capi.MEM_KINDS[st['mem_kind']]} "
              f"{n/secs:.3e} nodes/s props/node={st['num_deductions']/max(1,n):.0f} {st['num_deductions']/secs:.3e} props/s", flush=True)
