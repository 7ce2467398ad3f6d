#!/usr/bin/env python3
"""Write a MiniZinc solver configuration (.msc) for this build's `turbo` executable.

usage: scripts/make_msc.py [--mznlib DIR] [--out turbo.mi355x.msc]
The keys follow the reference's benchmarks/minizinc/turbo.gpu.release.msc; `mznlib` must point to a MiniZinc
library directory with the solver's redefinitions (the reference ships one under benchmarks/minizinc/mzn-lib).
MiniZinc itself is not needed to solve .fzn / .xml files with `turbo`.
"""
import argparse, json, os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--mznlib", default="")
ap.add_argument("--out", default=os.path.join(ROOT, "turbo.mi355x.msc"))
a = ap.parse_args()
cfg = {
    "executable": os.path.join(ROOT, "turbo_amd", "bin", "turbo"),
    "id": "turbo.mi355x", "name": "Turbo (MI355X, HIP)", "version": "1.3.0",
    "isGUIApplication": False, "mznlibVersion": 1,
    "needsMznExecutable": False, "needsPathsFile": False, "needsSolns2Out": True, "needsStdlibDir": False,
    "stdFlags": ["-a", "-n", "-p", "-s", "-v", "-f", "-t"],
    "supportsFzn": True, "supportsMzn": False, "supportsNL": False,
    "extraFlags": [
        ["-sub", "number of EPS subproblems (value of 10 generates 2^10 subproblems)", "int", "-1"],
        ["-subfactor", "B workgroups generate B * factor subproblems", "int", "300"],
        ["-or", "number of workgroups", "int", "0"],
        ["-arch", "architecture to use (gpu | barebones)", "string", "gpu"],
        ["-fp", "fixpoint (ac1 | wac1 | event)", "string", "wac1"],
        ["-wac1_threshold", "propagator count under which WAC1 falls back to AC1", "int", "0"],
        ["-gpus", "number of GPUs of the node", "int", "1"],
        ["-timeout", "soft timeout in ms (statistics are still printed)", "int", "0"],
        ["-globalmem", "keep every store in global memory", "bool", "false"],
        ["-disable_simplify", "disable the network simplifier", "bool", "false"],
        ["-disable_network_analysis", "do not print the network statistics", "bool", "false"],
        ["-cutnodes", "node budget per workgroup", "int", "0"],
        ["-eps_var_order", "variable ordering during the diving phase", "string", "default"],
        ["-eps_value_order", "value ordering during the diving phase", "string", "default"],
        ["-seed", "random seed", "int", "0"],
    ],
}
if a.mznlib:
    cfg["mznlib"] = os.path.abspath(a.mznlib)
with open(a.out, "w") as f:
    json.dump(cfg, f, indent=4)
print(a.out)
