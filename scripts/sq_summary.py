#!/usr/bin/env python3
"""profiles/<tag>_sq_summary.json from two scripts/pmc_quick.sh outputs (WAC1 and event runs of bench.py).

usage: scripts/sq_summary.py <tag> <pmc_quick output of the wac1 run> <pmc_quick output of the event run>
busy = counter / (1024 SIMDs x launch time x quad-cycle rate); the rate comes from SQ_WAVE_CYCLES of the WAC1 run,
whose 4096 waves are resident for the whole launch.
"""
import json, os, re, sys

tag, wac1_path, event_path = sys.argv[1:4]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(path):
    d = {}
    for line in open(path):
        m = re.match(r"(SQ_\w+)\s+([0-9.e+]+)", line)
        if m:
            d[m.group(1)] = float(m.group(2))
        m = re.search(r'"avg_launch_ms": ([0-9.]+)', line)
        if m:
            d["avg_launch_ms"] = float(m.group(1))
        m = re.search(r'"value": ([0-9.]+)', line)
        if m:
            d["props_per_s"] = float(m.group(1))
    return d


w, e = parse(wac1_path), parse(event_path)
qps = w["SQ_WAVE_CYCLES"] / 4096.0 / (w["avg_launch_ms"] * 1e-3)  # quad-cycles per second and wave slot
out = {}
for name, d in (("wac1", w), ("event", e)):
    secs = d["avg_launch_ms"] * 1e-3
    cap = 1024 * secs * qps
    wave_props = d["props_per_s"] * secs / 64
    out[name] = {"avg_launch_ms": d["avg_launch_ms"], "valu_busy": d["SQ_ACTIVE_INST_VALU"] / cap, "salu_busy": d["SQ_ACTIVE_INST_SCA"] / cap,
                 "lds_busy": d["SQ_ACTIVE_INST_LDS"] / cap, "valu_insts_per_64_propagations": d["SQ_INSTS_VALU"] / wave_props,
                 "salu_insts_per_64_propagations": d["SQ_INSTS_SALU"] / wave_props,
                 "counters": {k: v for k, v in d.items() if k.startswith("SQ_")}}
rec = {"note": "rocprofv3 --pmc (scripts/pmc_quick.sh) on bench.py --steps 2 --warmup 1 (--fixpoint event for the second); per-launch sums of "
               "tb::solve_kernel; SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES are in quad-cycles; busy = counter / (1024 SIMDs x launch time x "
               f"{qps:.3e} quad-cycles/s)", "kernels": out}
json.dump(rec, open(os.path.join(ROOT, "profiles", f"{tag}_sq_summary.json"), "w"), indent=1)
for k, v in out.items():
    print(k, {a: round(b, 3) for a, b in v.items() if not isinstance(b, dict)})
