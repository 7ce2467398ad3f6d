#!/usr/bin/env python3
"""Which spilled scalars cost the most per node: lane spills / reloads of a marker build (kernels.hpp: TB_REGION) weighted by the region census of the tuning build.
usage: isa_hot_spills.py build/h_regions.s gpurun_out/r04_region_census.json [regions to list]"""
import collections, json, re, sys
lines = open(sys.argv[1]).read().split("\n")
ce = json.load(open(sys.argv[2])); n = ce["nodes"]
start = next(i for i, l in enumerate(lines) if (m := re.match(r"^(_Z\w+):", l)) and "solve_kernel" in m.group(1))
cur, loc, defs, lane_def, per = 0, None, {}, {}, collections.defaultdict(list)
spillv = set()
for i in range(start + 1, len(lines)):
    s = lines[i].strip()
    if s.startswith(".Lfunc_end"): break
    if s.startswith("v_writelane_b32"): spillv.add(s.split()[1].rstrip(","))
for i in range(start + 1, len(lines)):
    s = lines[i].strip()
    if s.startswith(".Lfunc_end"): break
    m = re.match(r";\s*TBREGION\s+(\d+)", s)
    if m: cur = int(m.group(1)); continue
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if m:
        if int(m.group(2)): loc = int(m.group(2))
        continue
    if not s or s[0] in ";." or s.endswith(":"): continue
    p = s.replace(",", " ").split()
    if s.startswith("v_writelane_b32"):
        lane_def[(p[1], p[3])] = defs.get(p[2], ("?", None))
        per[cur].append(("spill ", p[1], p[3], defs.get(p[2], ("?", None))))
    elif s.startswith("v_readlane_b32") and p[2] in spillv:
        per[cur].append(("reload", p[2], p[3], lane_def.get((p[2], p[3]), ("?", None))))
    elif s.startswith("scratch_"):
        per[cur].append(("scratch", p[0], "", ("", loc)))
    elif s.startswith("s_") and len(p) > 1:
        m2 = re.match(r"s\[(\d+):(\d+)\]", p[1])
        for r in ([f"s{k}" for k in range(int(m2.group(1)), int(m2.group(2)) + 1)] if m2 else [p[1]] if re.match(r"s\d+$", p[1]) else []):
            defs[r] = (s[:60], loc)
rows = sorted(((ce["passes"].get(str(r), 0) / n * len(v), r, len(v)) for r, v in per.items()), reverse=True)
print("total per node (upper bound): %.0f" % sum(x[0] for x in rows))
print("per region (ops per node, region, static ops):", [(round(a), r, c) for a, r, c in rows[:10]])
for reg in [int(a) for a in sys.argv[3:]] or [r for _, r, _ in rows[:4]]:
    print("== region", reg)
    for x in per[reg]: print("  ", x[0], x[1], x[2], "<=", x[3][0][:58], "@", x[3][1])
