#!/usr/bin/env python3
"""Summary of a rocprofv3 PC-sampling CSV (scripts/r05_pc_sampling.sh): where the waves of the search kernel ARE, sample by sample -- the budget in wave-time
instead of instruction counts (the static x dynamic instruction budget left 1.4x / 2.1x unattributed).

    python3 scripts/pcs_summary.py <pc_sampling csv> [top]

Every sample is one wave's PC at a sampling instant.  A sample is attributed to (i) the source line its instruction was compiled from (line tables: the library
must be built with -gline-tables-only; inlined code carries the line of the function it came from), (ii) the function of kernels.hpp / propagators.hpp that line
lies in, (iii) the last TB_REGION marker before the line within that function (scripts/region_budget.py: NAMES), (iv) the kind of instruction."""
import collections
import csv
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
csv.field_size_limit(1 << 30)


def source_map(fname):
    """line -> (function, region) for one source file"""
    out, fn, reg = {}, "?", None
    try:
        lines = open(fname).read().split("\n")
    except OSError:
        return out
    depth = 0
    for i, l in enumerate(lines, 1):
        if depth == 0:
            m = re.match(r"^\s*(?:template\s*<[^>]*>\s*)?(?:__device__|__global__|static|inline)[^;{]*?\b([A-Za-z_]\w*)\s*\(", l)
            if m and m.group(1) not in ("__launch_bounds__", "__attribute__", "aligned"):
                fn, reg = m.group(1), None
            m2 = re.match(r"^\s*__global__.*\b(solve_kernel\w*|propagate_kernel\w*)\s*\(", l)
            if m2:
                fn, reg = m2.group(1), None
        m = re.search(r"TB_REGION\((\d+)\)", l)
        if m and "#define" not in l:
            reg = int(m.group(1))
        out[i] = (fn, reg)
        if re.match(r"^\s*namespace\b.*\{\s*$", l) or re.match(r"^\}\s*//\s*namespace", l):
            continue  # (the namespace's braces are not a nesting level)
        depth += l.count("{") - l.count("}")
        if depth < 0:
            depth = 0
    return out


maps = {}


def where(comment):
    m = re.search(r"([\w./-]+\.(?:hpp|hip|h)):(\d+)", comment or "")
    if not m:
        return "?", 0, "?", None
    f, ln = os.path.basename(m.group(1)), int(m.group(2))
    if f not in maps:
        maps[f] = source_map(os.path.join(ROOT, "turbo_amd", "csrc", "hip", f))
    fn, reg = maps[f].get(ln, ("?", None))
    return f, ln, fn, reg


def kind(ins):
    op = (ins or "?").split()[0] if ins else "?"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt " + (" ".join(ins.split()[1:]) if len(ins.split()) > 1 else "")
    if op.startswith("s_barrier"):
        return "s_barrier"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane moves (v_readlane / v_writelane / v_readfirstlane)"
    if op.startswith("v_"):
        return "VALU"
    if op.startswith(("s_load", "s_buffer_load")):
        return "scalar loads"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branches"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vector memory"
    return op


try:
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    NAMES = {}
    src = open(os.path.join(ROOT, "scripts", "region_budget.py")).read()
    for m in re.finditer(r"(\d+): \"([^\"]+)\"", src):
        NAMES[int(m.group(1))] = m.group(2)
except OSError:
    NAMES = {}

rd = csv.DictReader(open(path, newline=""))
cols = {c.lower(): c for c in rd.fieldnames}
c_ins = cols.get("instruction")
c_com = cols.get("instruction_comment")
print("columns:", rd.fieldnames)
n = 0
by_line, by_fn, by_reg, by_kind, by_file = (collections.Counter() for _ in range(5))
line_ins = collections.defaultdict(collections.Counter)
fn_kind = collections.defaultdict(collections.Counter)
for row in rd:
    ins = row.get(c_ins, "") if c_ins else ""
    f, ln, fn, reg = where(row.get(c_com, "") if c_com else "")
    n += 1
    k = kind(ins)
    by_kind[k.split(" ")[0] if k.startswith("s_waitcnt") else k] += 1
    by_file[f] += 1
    by_fn[fn] += 1
    fn_kind[fn][k.split(" ")[0] if k.startswith("s_waitcnt") else k] += 1
    by_reg[(fn, reg)] += 1
    by_line[(f, ln)] += 1
    line_ins[(f, ln)][ins.strip()[:60]] += 1
print(f"samples: {n}")
if not n:
    sys.exit(0)


def pct(c):
    return f"{100.0 * c / n:5.1f} %"


print("\n== by kind of instruction the wave was at")
for k, c in by_kind.most_common(20):
    print(f"  {pct(c)}  {k}")
print("\n== by source file")
for k, c in by_file.most_common(8):
    print(f"  {pct(c)}  {k}")
print(f"\n== by function (top {top}); in brackets: the three commonest kinds")
for k, c in by_fn.most_common(top):
    kk = ", ".join(f"{a} {100.0 * b / c:.0f} %" for a, b in fn_kind[k].most_common(3))
    print(f"  {pct(c)}  {k}   [{kk}]")
print(f"\n== by function and last TB_REGION marker before the line (top {top})")
for (fn, reg), c in by_reg.most_common(top):
    print(f"  {pct(c)}  {fn} / region {reg}: {NAMES.get(reg, '') if reg is not None else '(before the first marker)'}")
print(f"\n== by source line (top {top + 25}); the commonest instruction sampled there")
for (f, ln), c in by_line.most_common(top + 25):
    ins, ci = line_ins[(f, ln)].most_common(1)[0]
    fn = maps.get(f, {}).get(ln, ("?", None))[0]
    print(f"  {pct(c)}  {f}:{ln} ({fn})   {ins}  [{100.0 * ci / c:.0f} % of the line's samples]")
