#!/usr/bin/env python3
"""Static instruction census of one kernel by source line: compile the device code with -gline-tables-only -S and attribute every
instruction to the .loc in force (the innermost inlined frame).  Usage: isa_lines.py file.s mangled-substring [top]
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-atomic-optimizer-strategy=None -gline-tables-only -S --cuda-device-only -Iinclude -o /tmp/e.s turbo_amd/csrc/hip/engine.hip"""
import collections, re, sys
path, pat = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
files, inside, loc = {}, False, None
per = collections.defaultdict(lambda: collections.Counter())
with open(path) as f:
    for l in f:
        s = l.strip()
        m = re.match(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', s)
        if m: files[int(m.group(1))] = m.group(2); continue
        if not inside:
            m = re.match(r"^(_Z\w+):", l)
            if m and pat in m.group(1): inside = True
            continue
        if s.startswith(".Lfunc_end"): break
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m: loc = (files.get(int(m.group(1)), m.group(1)), int(m.group(2))); continue
        if not s or s[0] in ";." or s.endswith(":"): continue
        op = s.split()[0]
        kind = "valu" if op.startswith("v_") and not op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")) else \
               "lane" if op.startswith("v_") else "salu" if op.startswith("s_") else "ds" if op.startswith("ds_") else \
               "scratch" if op.startswith("scratch_") else "vmem" if op.startswith(("global_", "buffer_", "flat_")) else "other"
        per[loc][kind] += 1
tot = collections.Counter()
for c in per.values(): tot.update(c)
print("total", dict(tot))
rows = sorted(per.items(), key=lambda kv: -(kv[1]["valu"] + kv[1]["salu"]))[:top]
for (fn, ln), c in rows:
    print(f"{fn}:{ln:<6} valu={c['valu']:<5} salu={c['salu']:<5} lane={c['lane']:<4} ds={c['ds']:<4} vmem={c['vmem']:<4} scratch={c['scratch']}")
