#!/usr/bin/env python3
"""One budgeted event-fixpoint search with a tuning knob (tuning build), for instruction counting under rocprofv3 --pmc:
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU -d out -o p -- python3 scripts/valu_by_phase.py <knob bits, hex> [workload] [nodes]
Knob bits 8-15 select one phase executed twice, 0x1 repeats the successor marks, 0x4 the snapshot push, 0x8 adds one evaluation pass
per run (kernels.hpp: reps_of): the difference to knob 0 is that phase's instructions.  Driven by scripts/phase_budget.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import capi, preprocess
bits = int(sys.argv[1], 16) if len(sys.argv) > 1 else 0
fzn = {"wordpress7_500": "example_wordpress7_500.fzn", "accap_a3": "accap_a3.fzn", "trains15": "trains15.fzn"}[sys.argv[2] if len(sys.argv) > 2 else "wordpress7_500"]
nodes = int(sys.argv[3]) if len(sys.argv) > 3 else 12_000_000
_, tcn, _ = preprocess.load_fzn_simplified(os.path.join(ROOT, "benchmarks", fzn))
# optional 4th argument proof:<bound>:<power>: the whole search that refutes objective <= bound (a constant constraint: the tree does not depend on timing -- what the
# instrumented and the counted run of scripts/r06_blocks.sh need to walk the SAME tree); the node budget is then ignored
if len(sys.argv) > 4 and sys.argv[4].startswith("proof:"):
    _, bound, power = sys.argv[4].split(":")
    cfg = capi.make_config(fixpoint=2, use_fixed_bound=1, fixed_bound=int(bound), subproblems_power=int(power), timeout_ms=600000, debug=bits)
else:
    cfg = capi.make_config(fixpoint=2, stop_after_n_nodes_total=nodes, timeout_ms=120000, debug=bits)
has, best, st = capi.solve(tcn, cfg)
print(f"bits={bits:#x} nodes={st['nodes']} fails={st['fails']} deductions={st['num_deductions']} kernel_ns={st['kernel_ns']} nodes/s={st['nodes'] / (st['kernel_ns'] * 1e-9):.4e}")
