"""Quick throughput probe (not the bench): GPU engine vs CPU oracle on the headline instances."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from turbo_amd import frontend, capi
from oracle import pyoracle

insts = sys.argv[1:] or ["example_wordpress7_500.fzn", "accap_a3.fzn", "trains15.fzn"]
for name in insts:
    t0 = time.time()
    tcn = frontend.load_fzn(os.path.join(ROOT, "benchmarks", name))
    print(f"== {name}: V={tcn.n_vars} P={tcn.n_props} strategies={tcn.n_strats} parse={time.time()-t0:.2f}s", flush=True)
    ops = {}
    for o in tcn.props["op"]:
        ops[int(o)] = ops.get(int(o), 0) + 1
    print("   ops:", {frontend.OP_NAMES[k]: v for k, v in sorted(ops.items())})
    for label, kw in [("wac1", dict(fixpoint=1)), ("event", dict(fixpoint=2))]:
        has, best, st = capi.solve(tcn, capi.make_config(timeout_ms=5000, **kw))
        secs = st["kernel_ns"] * 1e-9
        print(f"   GPU {label}: blocks={st['num_blocks']}x{st['threads_per_block']} mem={capi.MEM_KINDS[st['mem_kind']]} lds={st['shared_bytes']} d={st['subproblems_power']} "
              f"nodes={st['nodes']} ({st['nodes']/secs:.3e}/s) props={st['num_deductions']} ({st['num_deductions']/secs:.3e}/s) iters/node={st['fixpoint_iterations']/max(1,st['nodes']):.1f} "
              f"obj={tcn.objective_of(best) if has else None} exh={st['exhaustive']} t={secs:.2f}s fix%={st['timers_ns'][3]/max(1,st['cumulative_time_block_ns']):.2f} search%={st['timers_ns'][2]/max(1,st['cumulative_time_block_ns']):.2f}", flush=True)
    has, best, so = pyoracle.solve(tcn, timeout_ms=5000)
    print(f"   CPU oracle: nodes={so['nodes']} ({so['nodes']/so['solve_seconds']:.3e}/s) props={so['num_deductions']} ({so['num_deductions']/so['solve_seconds']:.3e}/s) iters/node={so['fixpoint_iterations']/max(1,so['nodes']):.1f} obj={tcn.objective_of(best) if has else None}", flush=True)
