#!/usr/bin/env python3
"""Slice runs per node of the event-driven fixpoint on the sequential DFS (one workgroup, 2^0 subproblems): the same node
population as tests/tools/event_profile.py simulates on the CPU.  Needs the tuning build (0x400000 counts slice runs):
TURBO_HIP_LIB=turbo_amd/lib/libturbo_hip_tuning.so python scripts/slice_runs_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import capi, preprocess
name = sys.argv[1] if len(sys.argv) > 1 else "example_wordpress7_500.fzn"
_, tcn, _ = preprocess.load_fzn_simplified(os.path.join(ROOT, "benchmarks", name))
for nodes in (400, 4000):
    for threads in (64, 128, 256, 1024):
        for power, blocks in ((0, 1), (19, 1536)):
            row = []
            for bits in (0, 0x400000):
                cfg = capi.make_config(fixpoint=2, or_nodes=blocks, subproblems_power=power, stop_after_n_nodes=nodes, threads_per_block=threads, timeout_ms=120000, debug=bits)
                has, best, st = capi.solve(tcn, cfg)
                row.append(st["num_deductions"] / 64.0 / max(1, st["nodes"]))
                secs = st["kernel_ns"] * 1e-9
            print(f"{name} nodes/wg={nodes} threads={threads} workgroups={st['num_blocks']} sub=2^{power}: wave iterations/node {row[0]:.1f}, slice runs/node {row[1]:.1f}, "
                  f"{st['nodes'] / secs:.3e} nodes/s mem={capi.MEM_KINDS[st['mem_kind']]}", flush=True)
