#!/usr/bin/env python3
"""Per-kernel instruction census of a `hipcc -save-temps` .s file: SGPR spills through VGPR lanes
(v_readlane / v_writelane), scratch, register counts.  Usage: isa_stats.py file.s [substring of the mangled name]"""
import re
import subprocess
import sys

path = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cur, stats = None, {}
for line in open(path):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = m.group(1)
        stats[cur] = {"insts": 0, "readlane": 0, "writelane": 0, "scratch": 0, "valu": 0, "salu": 0, "ds": 0, "vmem": 0}
        continue
    if cur is None:
        continue
    s = line.strip()
    if s.startswith(".") or s.startswith(";") or not s:
        m = re.match(r";\s*(NumSgprs|NumVgprs|ScratchSize|Occupancy|LDSByteSize|NumAgprs|TotalNumVgprs)\s*:\s*(\d+)", s)
        if m:
            stats[cur][m.group(1)] = int(m.group(2))
        continue
    op = s.split()[0]
    st = stats[cur]
    st["insts"] += 1
    if op.startswith("v_readlane"):
        st["readlane"] += 1
    elif op.startswith("v_writelane"):
        st["writelane"] += 1
    if op.startswith("scratch_"):
        st["scratch"] += 1
    if op.startswith("v_"):
        st["valu"] += 1
    elif op.startswith("s_"):
        st["salu"] += 1
    elif op.startswith("ds_"):
        st["ds"] += 1
    elif op.startswith(("global_", "buffer_", "flat_")):
        st["vmem"] += 1
for k, v in stats.items():
    if pat and pat not in k:
        continue
    if v["insts"] < 50:
        continue
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    print(name[:100])
    print("   ", v)
