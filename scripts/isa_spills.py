#!/usr/bin/env python3
"""Which scalar values a kernel spills into VGPR lanes / scratch and where it reloads them (static, from a -gline-tables-only -S listing):
isa_spills.py file.s mangled-substring [first_line last_line]  -- the optional source-line range delimits a region (e.g. the round loop of
fixpoint_event) whose spill traffic is counted separately."""
import collections, re, sys
path, pat = sys.argv[1], sys.argv[2]
lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, 0)
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if (m := re.match(r"^(_Z\w+):", l)) and pat in m.group(1))
seq, loc, lastnz = [], None, None
for i in range(start + 1, len(lines)):
    s = lines[i].strip()
    if s.startswith(".Lfunc_end"): break
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if m:
        loc = int(m.group(2))
        if loc: lastnz = loc
        continue
    if not s or s[0] in ";." or s.endswith(":"): continue
    seq.append((i, loc, lastnz, s))
spillv = collections.Counter(s.split()[1].rstrip(",") for _, _, _, s in seq if s.startswith("v_writelane_b32"))
print("spill VGPRs:", dict(spillv))
defs, rows = {}, []
for i, loc, lnz, s in seq:
    if s.startswith("v_writelane_b32"):
        p = s.replace(",", " ").split()
        rows.append((p[1], int(p[3]), defs.get(p[2], ("?", None))))
    elif s.startswith("s_"):
        p = s.replace(",", " ").split()
        if len(p) > 1:
            m = re.match(r"s\[(\d+):(\d+)\]", p[1])
            for r in ([f"s{k}" for k in range(int(m.group(1)), int(m.group(2)) + 1)] if m else [p[1]] if re.match(r"s\d+$", p[1]) else []):
                defs[r] = (s, lnz)
by_line = collections.Counter(d[1] for _, _, d in rows)
print("lane spills by the source line that defined the value:", sorted(by_line.items(), key=lambda kv: -kv[1])[:25])
if hi:
    idx = [k for k, (_, loc, _, _) in enumerate(seq) if loc is not None and lo <= loc <= hi]
    first, last = idx[0], idx[-1]
    c = collections.Counter()
    for _, loc, lnz, s in seq[first:last + 1]:
        op = s.split()[0]
        if op.startswith("v_readlane"): c["reload" if s.split()[2].rstrip(",") in spillv else "readlane (not a spill)"] += 1
        elif op.startswith("v_writelane"): c["spill"] += 1
        elif op.startswith("scratch_"): c["scratch"] += 1
        elif op.startswith("v_"): c["valu"] += 1
        elif op.startswith("s_"): c["salu"] += 1
    print(f"region of source lines {lo}-{hi}: instructions {last - first + 1}", dict(c))
