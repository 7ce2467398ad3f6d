#!/usr/bin/env python3
"""Basic-block census of one kernel in a -save-temps .s file: per label, instruction mix and spill traffic.
Usage: isa_blocks.py file.s mangled-substring"""
import re, sys
path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = None
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if m and pat in m.group(1):
        start = i; name = m.group(1); break
assert start is not None
blocks, cur = [], {"label": "entry", "n": 0, "rl": 0, "wl": 0, "ds": 0, "vm": 0, "valu": 0, "salu": 0, "br": [], "line": start}
for i in range(start + 1, len(lines)):
    s = lines[i].strip()
    if s.startswith(".Lfunc_end"): break
    m = re.match(r"^(\.LBB\d+_\d+):", s)
    if m:
        blocks.append(cur)
        cur = {"label": m.group(1), "n": 0, "rl": 0, "wl": 0, "ds": 0, "vm": 0, "valu": 0, "salu": 0, "br": [], "line": i}
        continue
    if not s or s.startswith(";") or s.startswith("."): continue
    op = s.split()[0]
    cur["n"] += 1
    if op.startswith("v_readlane"): cur["rl"] += 1
    elif op.startswith("v_writelane"): cur["wl"] += 1
    elif op.startswith("ds_"): cur["ds"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): cur["vm"] += 1
    elif op.startswith("v_"): cur["valu"] += 1
    elif op.startswith("s_"):
        cur["salu"] += 1
        if op.startswith(("s_cbranch", "s_branch")): cur["br"].append(s.split()[-1])
blocks.append(cur)
print(name, len(blocks), "blocks")
for b in blocks:
    print(f"{b['label']:>12} L{b['line']:<8} n={b['n']:<4} valu={b['valu']:<4} salu={b['salu']:<4} ds={b['ds']:<3} vm={b['vm']:<3} rl={b['rl']:<3} wl={b['wl']:<3} -> {','.join(b['br'])}")
