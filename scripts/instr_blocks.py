#!/usr/bin/env python3
"""Basic-block execution counts of the PRODUCTION event kernels, without PC sampling or thread trace (neither is available on the pool).

The budget r04 / r05 tried to close (static instruction counts between source markers x marker passes) was off by 1.4x / 2.1x: the compiler moves cold blocks away
from their markers and a region holds branches a wave does not take.  This tool counts what is executed, at the granularity the hardware counters use:

  build    scripts/instr_blocks.py build [out.so]
           compiles turbo_amd/csrc/hip/unit_1.hip (the 128-thread event kernels: wordpress7_500, accap_a3, trains15) exactly as `make hip` does, with line tables,
           keeps the device assembly, and REWRITES it: at the head of every straight-line segment (a label or the instruction after a branch ... the next branch) of the
           selected kernels one `s_atomic_add <one>, <counters>, 4 * segment` -- a scalar-memory atomic: no vector register, no exec mask, no VCC; it uses three SGPRs
           above the ones the kernel allocates.  The counter array comes from the kernel arguments (DevProblem::blk_counts), one copy per XCD (HW_REG_XCC_ID) because the
           L2s of different XCDs are not coherent for such atomics.  Then re-runs the rest of hipcc's pipeline (assembler, lld, bundler, host compile) and links the
           instrumented unit with the production objects of the other units: turbo_amd/lib/libturbo_hip_blocks.so + build/instr/segments.json.
           Everything else -- register allocation, scheduling, spills, block layout -- is the production kernel's, instruction for instruction.
  run      TURBO_HIP_LIB=.../libturbo_hip_blocks.so TB_BLOCK_COUNTS=counts.bin  <any driver of ONE search>     (engine.hip: tb_session_finish writes the counts)
  table    scripts/instr_blocks.py table segments.json counts.bin <kernel substring> [nodes] [SQ_INSTS_VALU SQ_INSTS_SALU measured on the production library, same search]
           dynamic instruction counts per source region (kernels.hpp: the TB_REGION points; helpers inlined into a region count for it), per class of instruction,
           their sum against the hardware counters, and the hottest segments.

What the hardware counters count (calibrated by the sum): SQ_INSTS_VALU = v_* (v_readlane / v_writelane included, memory instructions not); SQ_INSTS_SALU = s_* ALU and
moves (not s_waitcnt / s_nop / s_barrier / s_sleep / branches / s_load).
"""
import collections
import json
import os
import re
import struct
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORK = os.path.join(ROOT, "build", "instr")
HIPFLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-parameter", "-Wno-bitwise-instead-of-logical", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]
STRIDE_LOG2 = 16  # device_types.hpp: BLK_COUNT_STRIDE

BRANCH = ("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm", "s_call")
MISC = ("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_sethalt", "s_setprio", "s_dcache", "s_icache", "s_endpgm", "s_trap", "s_sendmsg", "s_ttrace", "s_inst_prefetch", "s_clause", "s_code_end", "s_setkill", "s_incperflevel", "s_decperflevel")
SMEM = ("s_load", "s_buffer_load", "s_store", "s_atomic", "s_scratch", "s_memtime", "s_memrealtime", "s_buffer_store", "s_buffer_atomic")


def classify(op: str) -> str:
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith(BRANCH):
        return "branch"
    if op.startswith(SMEM):
        return "smem"
    if op.startswith(MISC):
        return "misc"
    if op.startswith("s_"):
        return "salu"
    return "other"


def offset_of_blk_counts() -> int:
    src = os.path.join(WORK, "off.hip")
    open(src, "w").write('#include "%s/turbo_amd/csrc/hip/device_types.hpp"\n#include <cstdio>\nint main(){ printf("%%zu\\n", offsetof(tb::DevProblem, blk_counts)); }\n' % ROOT)
    subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "--offload-arch=gfx950", "-w", src, "-o", os.path.join(WORK, "off")], check=True)
    return int(subprocess.run([os.path.join(WORK, "off")], capture_output=True, text=True, check=True).stdout.strip())


def instrument(lines, want, karg_off):
    """Returns (new lines, segments): `want(kernel_name)` selects the kernels to rewrite."""
    out, segs = [], []
    i, n = 0, len(lines)
    fn_re = re.compile(r"^(_Z\w+):\s*(;.*)?$")
    while i < n:
        m = fn_re.match(lines[i])
        if not m or "solve_kernel" not in m.group(1) or not want(m.group(1)):
            out.append(lines[i]); i += 1
            continue
        name = m.group(1)
        # the extent of the function and its descriptor
        j = i + 1
        while j < n and not lines[j].startswith(".Lfunc_end"):
            j += 1
        k = next(t for t in range(i, j) if ".amdhsa_next_free_sgpr" in lines[t])  # (the kernel descriptor sits between the code and .Lfunc_end)
        free = int(lines[k].split()[-1])
        base = (free + 1) & ~1
        assert base + 3 <= 102, (name, free)
        one = base + 2
        out.append(lines[i])
        # entry: counters base (+ XCD copy) and the constant one; s[0:1] = kernarg segment pointer (user SGPRs: 2)
        out += [f"\ts_load_dwordx2 s[{base}:{base + 1}], s[0:1], {hex(karg_off)}",
                f"\ts_getreg_b32 s{one}, hwreg(20, 0, 4)",
                "\ts_waitcnt lgkmcnt(0)",
                f"\ts_lshl_b32 s{one}, s{one}, {STRIDE_LOG2}",
                f"\ts_add_u32 s{base}, s{base}, s{one}",
                f"\ts_addc_u32 s{base + 1}, s{base + 1}, 0",
                f"\ts_mov_b32 s{one}, 1"]
        kernel_first = len(segs)
        cur = None          # the open segment
        line_ctx = 0        # last .loc line seen (kernels.hpp only)
        file_ids = {}
        start_new = True    # the next instruction opens a segment
        for t in range(i + 1, j):
            raw = lines[t]
            s = raw.strip()
            mf = re.match(r"\.file\s+(\d+)\s+(.*)$", s)
            if mf:
                file_ids[int(mf.group(1))] = mf.group(2)
            ml = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
            if ml:
                # the assembly comment behind a .loc holds the whole inline stack: `; file:line:col @[ caller:line:col @[ ... ] ]` -- kept as "file:line>file:line>..." (innermost first)
                mc = re.search(r";\s*(\S.*)$", s)
                frames = []
                if mc:
                    for fr in mc.group(1).replace("]", " ").split("@["):
                        mm = re.match(r"\s*(\S+?):(\d+)(?::\d+)?\s*$", fr.strip())
                        if mm:
                            frames.append(f"{os.path.basename(mm.group(1))}:{mm.group(2)}")
                line_ctx = (int(ml.group(1)), int(ml.group(2)), ">".join(frames))
                out.append(raw)
                continue
            if t == k:
                out.append(raw.replace(str(free), str(base + 3)))
                continue
            if re.match(r"^\.?L?[A-Za-z_0-9.$]+:", s) and not s.startswith(";"):  # a label: only basic-block labels are branch targets (.Ltmp / .Lfunc_begin are debug-info marks)
                if re.match(r"^\.LBB\d+_\d+:", s):
                    start_new = True
                out.append(raw)
                continue
            if not s or s[0] in ";." or s.startswith("//"):
                out.append(raw)
                continue
            op = s.split()[0]
            if start_new:
                sid = len(segs) - kernel_first
                assert 4 * sid < (1 << STRIDE_LOG2)
                cur = {"kernel": name, "id": sid, "counts": collections.Counter(), "locs": collections.Counter(), "first_loc": line_ctx[2] if line_ctx else "", "text": []}
                segs.append(cur)
                # The counting atomic is fenced: it is never outstanding together with the kernel's own LDS / scalar-memory operations.  Unfenced (TB_INSTR_FENCE=0), one
                # 12 M-node search of wordpress7_500 in five took a wrong decision (a variable-selection reduction over ds_bpermute consumed a stale value) -- the returnless
                # scalar atomics interfere with the lgkmcnt waits the compiler counts for in-order LDS returns; fenced, 0 of 24 runs did.
                if os.environ.get("TB_INSTR_FENCE", "1") != "0":
                    out.append("\ts_waitcnt lgkmcnt(0)")
                out.append(f"\ts_atomic_add s{one}, s[{base}:{base + 1}], {hex(4 * sid)}")
                if os.environ.get("TB_INSTR_FENCE", "1") != "0":
                    out.append("\ts_waitcnt lgkmcnt(0)")
                start_new = False
            c = classify(op)
            cur["counts"][c] += 1
            cur["locs"][(line_ctx[2] or f"#{line_ctx[0]}:{line_ctx[1]}") if line_ctx else "?:0", c] += 1
            if len(cur["text"]) < 400:
                cur["text"].append(s.split(";")[0].strip())
            out.append(raw)
            if op.startswith(BRANCH):
                start_new = True
        i = j
    return out, segs


def relax_branches(path, asm_cmd):
    """The rewritten kernels are ~40 % longer, and a few of their branches no longer reach their targets (s_cbranch / s_branch: +-32 K dwords).  The compiler's own
    relaxation (s_getpc + s_setpc) needs a free SGPR pair at the branch; here a branch that does not reach goes through an ISLAND instead: one `s_branch target` placed at
    a point no instruction falls into (right behind an unconditional branch), about half way between the two.  An island is not a segment: it does not exist in the
    production kernel and is not counted.  Repeated until the assembler is content."""
    for rnd in range(8):
        r = subprocess.run(asm_cmd, capture_output=True, text=True)
        errs = [int(m.group(1)) for m in re.finditer(r":(\d+):\d+: error: branch size exceeds simm16", r.stderr)]
        if r.returncode == 0:
            return
        assert errs, r.stderr[-3000:]
        lines = open(path).read().split("\n")
        label_line = {}
        for t, l in enumerate(lines):
            m = re.match(r"^(\.LBB\d+_\d+|\.Lisl_\d+):", l)
            if m:
                label_line[m.group(1)] = t
        safe = [t + 1 for t, l in enumerate(lines) if l.strip().split(" ")[0].split("\t")[0] in ("s_branch", "s_endpgm", "s_setpc_b64")]  # insert BEFORE line safe[k]
        import bisect
        inserts = collections.defaultdict(list)
        n_isl = sum(1 for l in lines if l.startswith(".Lisl_"))
        for ln in errs:
            t = ln - 1
            parts = lines[t].split()
            target = parts[1]
            mid = (t + label_line[target]) // 2
            k = bisect.bisect_left(safe, mid)
            pos = safe[min(max(k, 0), len(safe) - 1)]
            name = f".Lisl_{n_isl}"; n_isl += 1
            inserts[pos].append(f"{name}:\n\ts_branch {target}")
            lines[t] = lines[t].replace(target, name)
        out = []
        for t, l in enumerate(lines):
            for isl in inserts.get(t, []):
                out.append(isl)
            out.append(l)
        open(path, "w").write("\n".join(out))
        print(f"  branch relaxation round {rnd}: {len(errs)} branches through islands")
    raise RuntimeError("branch relaxation did not converge")


def cmd_build(out_so):
    os.makedirs(WORK, exist_ok=True)
    os.chdir(WORK)
    src = os.path.join(ROOT, "turbo_amd", "csrc", "hip", "unit_1.hip")
    log = subprocess.run(["/opt/rocm/bin/hipcc"] + HIPFLAGS + os.environ.get("TB_INSTR_FLAGS", "").split() + ["-gline-tables-only", "-c", "-save-temps", "-v", "-o", "unit_1.o", src], capture_output=True, text=True)
    assert log.returncode == 0, log.stderr[-3000:]
    steps = [l.strip() for l in log.stderr.splitlines() if l.startswith(' "')]
    dev_s = "unit_1-hip-amdgcn-amd-amdhsa-gfx950.s"
    lines = open(dev_s).read().split("\n")
    files = {}
    for l in lines:
        mf = re.match(r"\s*\.file\s+(\d+)\s+(.*)$", l)
        if mf:
            files[int(mf.group(1))] = mf.group(2).replace('"', "")
    sel = os.environ.get("TB_INSTR_KERNELS", "")  # substring filter on the mangled name ("" = every solve_kernel of the unit)
    new, segs = instrument(lines, lambda nm: sel in nm, offset_of_blk_counts())
    os.rename(dev_s, dev_s + ".orig")
    open(dev_s, "w").write("\n".join(new))
    import shlex
    asm_cmd = shlex.split(next(s for s in steps if "-cc1as" in s and "amdgcn" in s))
    relax_branches(dev_s, asm_cmd)
    # the rest of the pipeline, as hipcc ran it: device assembler, lld, bundler, host compile (which embeds the new bundle), host assembler
    import shlex
    replay = [s for s in steps if ("-cc1as" in s and "amdgcn" in s) or "lld" in s.split()[0] or "clang-offload-bundler" in s.split()[0]]
    host = [s for s in steps if re.search(r"(?<![\w-])-triple x86_64-unknown-linux-gnu", s) and ("-emit-llvm-bc" in s or " -S " in s or "-cc1as" in s)]  # (not the device steps: -aux-triple x86_64)
    for s in replay + host:
        r = subprocess.run(shlex.split(s), capture_output=True, text=True)
        assert r.returncode == 0, (s[:200], r.stderr[-3000:])
    objs = [os.path.join(ROOT, "build", "obj", "hip", f) for f in ["engine.o"] + [f"unit_{u}.o" for u in range(2, 10)]]
    if os.environ.get("TB_INSTR_FLAGS"):  # (a variant that changes shared structures: its own host shim)
        eng = os.path.join(WORK, "engine_variant.o")
        subprocess.run(["/opt/rocm/bin/hipcc"] + HIPFLAGS + os.environ["TB_INSTR_FLAGS"].split() + ["-c", "-o", eng, os.path.join(ROOT, "turbo_amd", "csrc", "hip", "engine.hip")], check=True, capture_output=True)
        objs[0] = eng
    for o in objs:
        assert os.path.exists(o), f"{o}: run `make -j8 hip` first"
    subprocess.run(["/opt/rocm/bin/hipcc"] + HIPFLAGS + ["-shared", "-o", out_so, os.path.join(WORK, "unit_1.o")] + objs, check=True, capture_output=True)
    for sg in segs:
        sg["counts"] = dict(sg["counts"])
        sg["locs"] = [[k[0], k[1], v] for k, v in sg["locs"].items()]
    json.dump({"files": files, "segments": segs}, open(os.path.join(WORK, "segments.json"), "w"))
    json.dump({"files": files, "segments": segs}, open(os.path.join(os.path.dirname(out_so), "blocks_segments.json"), "w"))  # (beside the library: build/ does not travel to the GPU box)
    kernels = collections.Counter(s["kernel"] for s in segs)
    print(f"built {out_so}: {len(segs)} segments in {len(kernels)} kernels")
    for k, v in kernels.items():
        print(f"  {v:6d} segments  {k}")


def regions_of_source():
    """kernels.hpp line -> region id: a TB_REGION(n) statement opens region n for the lines that follow it, up to the next one (in line order)."""
    src = open(os.path.join(ROOT, "turbo_amd", "csrc", "hip", "kernels.hpp")).read().split("\n")
    marks = []
    for ln, l in enumerate(src, 1):
        for m in re.finditer(r"TB_REGION\((\d+)\)", l):
            if "#define" not in l:
                marks.append((ln, int(m.group(1))))
    return marks, src


def cmd_table(seg_path, counts_path, kernel_sub, nodes=None, m_valu=None, m_salu=None):
    import ast
    txt = open(os.path.join(ROOT, "scripts", "region_budget.py")).read()  # (the names of the TB_REGION points: a dict literal of that script, which is not importable)
    a = txt.index("NAMES = {") + len("NAMES = ")
    NAMES = ast.literal_eval(txt[a:txt.index("}", a) + 1])
    NAMES.update({66: "lean product run", 67: "conditional wake-up of an element constraint's implications (cond2)"})
    meta = json.load(open(seg_path))
    files = {int(k): v for k, v in meta["files"].items()}
    kfile = next((k for k, v in files.items() if v.endswith("kernels.hpp")), None)
    pfile = next((k for k, v in files.items() if v.endswith("propagators.hpp")), None)
    raw = open(counts_path, "rb").read()
    counts = struct.unpack(f"<{len(raw) // 8}Q", raw)
    segs = [s for s in meta["segments"] if kernel_sub in s["kernel"]]
    kernels = sorted(set(s["kernel"] for s in segs))
    assert len(kernels) == 1, f"'{kernel_sub}' selects {kernels}"
    marks, src = regions_of_source()
    psrc = open(os.path.join(ROOT, "turbo_amd", "csrc", "hip", "propagators.hpp")).read().split("\n")
    import bisect

    def functions_of(text):
        """[(first line, name)] of the top-level function definitions of a header (a definition starts at column 0 with template / __device__ / __global__ / static / inline)."""
        fns = []
        for ln, l in enumerate(text, 1):
            if not re.match(r"^(static\s+|inline\s+)?(__device__|__global__|__host__|template\s*<|constexpr\s+\w+\s+\w+\()", l):
                continue
            if fns and fns[-1][0] == ln - 1 and text[ln - 2].startswith("template"):
                start = fns.pop()[0]  # `template <...>` on the line above the declarator
            else:
                start = ln
            decl = " ".join(text[ln - 1:ln + 2])
            decl = re.sub(r"__launch_bounds__\s*\((?:[^()]|\([^()]*\))*\)", "", decl)
            names = [m.group(1) for m in re.finditer(r"(\w+)\s*\(", decl) if m.group(1) not in ("__launch_bounds__", "__attribute__", "address_space", "alignas", "aligned", "if", "while", "for", "sizeof", "decltype")]
            fns.append((start, names[0] if names else "?"))
        return fns
    kfns, pfns = functions_of(src), functions_of(psrc)
    kstarts, pstarts = [f[0] for f in kfns], [f[0] for f in pfns]
    fn_marks = collections.defaultdict(list)
    for l, rid in marks:
        k = bisect.bisect_right(kstarts, l) - 1
        fn_marks[kfns[k][1] if k >= 0 else "?"].append((l, rid))

    def frame_fn(frame):
        """(function, region or None) of one frame "file:line"."""
        name, ln = frame.rsplit(":", 1)
        ln = int(ln)
        if name == "kernels.hpp" and ln > 0:
            k = bisect.bisect_right(kstarts, ln) - 1
            fn = kfns[k][1] if k >= 0 else "?"
            ms = fn_marks.get(fn)
            if not ms:
                return fn, None
            r = ms[0][1]
            for l, rid in ms:
                if l <= ln:
                    r = rid
            return fn, r
        if name == "propagators.hpp" and ln > 0:
            k = bisect.bisect_right(pstarts, ln) - 1
            return "propagators.hpp: " + (pfns[k][1] if k >= 0 else "?"), None
        return (name + " (line 0: compiler-generated)" if ln == 0 else name), None

    def where(loc):
        """loc = "file:line>file:line>..." (innermost first).  Returns (innermost function, region): the region is that of the first frame, going outwards, that lies in a
        function with TB_REGION points -- the call site the instruction was inlined into, taken from the compiler's inline stack (exact, no layout heuristics)."""
        frames = [f for f in loc.lstrip("#").split(">") if f]
        if not frames:
            return "?", None
        fn, reg = frame_fn(frames[0])
        for fr in frames:
            f2, r2 = frame_fn(fr)
            if r2 is not None:
                reg = r2
                break
        return fn, reg

    CLS = ("valu", "lane", "salu", "lds", "vmem", "scratch", "smem", "branch", "misc", "other")
    per_region = collections.defaultdict(collections.Counter)
    per_reg_only = collections.defaultdict(collections.Counter)
    per_line = collections.defaultdict(collections.Counter)
    tot = collections.Counter()
    seg_rows = []
    for s in segs:
        c = counts[s["id"]]
        for loc, cl, v in s["locs"]:
            fn, reg = where(loc)
            per_region[(fn, reg)][cl] += v * c
            per_reg_only[reg][cl] += v * c
            if c:
                f0 = loc.lstrip("#").split(">")[0]
                per_line[(f0.rsplit(":", 1)[0], int(f0.rsplit(":", 1)[1]) if ":" in f0 else 0)][cl] += v * c
            tot[cl] += v * c
        seg_rows.append((c * sum(s["counts"].values()), c, s))
    nodes = float(nodes) if nodes else None
    per = (lambda x: x / nodes) if nodes else (lambda x: x)
    out = {"kernel": kernels[0], "nodes": nodes, "unit": "wave-instructions per node" if nodes else "wave-instructions",
           "sum": {k: round(per(tot[k]), 1) for k in CLS if tot[k]},
           "what_the_counters_count": "SQ_INSTS_VALU = valu + lane; SQ_INSTS_SALU = salu (+ branch + misc when the sum says so, see `measured`)"}
    if m_valu:
        mv, ms = float(m_valu), float(m_salu)
        out["measured"] = {"SQ_INSTS_VALU_per_node": mv, "SQ_INSTS_SALU_per_node": ms,
                           "valu_plus_lane_over_measured": round(per(tot["valu"] + tot["lane"]) / mv, 4),
                           "salu_over_measured": round(per(tot["salu"]) / ms, 4),
                           "salu_plus_branch_over_measured": round(per(tot["salu"] + tot["branch"]) / ms, 4),
                           "salu_plus_branch_misc_over_measured": round(per(tot["salu"] + tot["branch"] + tot["misc"]) / ms, 4)}
    rows = []
    for (fn, reg), cnt in per_region.items():
        rows.append({"function": fn, "region": reg, "what": NAMES.get(reg, "") if reg is not None else "(the function's own instructions, wherever it was inlined)",
                     **{k: round(per(cnt[k]), 1) for k in CLS if cnt[k]},
                     "valu_share": round((cnt["valu"] + cnt["lane"]) / max(1, tot["valu"] + tot["lane"]), 4), "salu_share": round(cnt["salu"] / max(1, tot["salu"]), 4)})
    rows.sort(key=lambda r: -(r.get("valu", 0) + r.get("lane", 0) + r.get("salu", 0)))
    by_region = []
    for reg, cnt in per_reg_only.items():
        by_region.append({"region": reg, "what": NAMES.get(reg, "?") if reg is not None else "outside every marked function", **{k: round(per(cnt[k]), 1) for k in CLS if cnt[k]},
                          "valu_share": round((cnt["valu"] + cnt["lane"]) / max(1, tot["valu"] + tot["lane"]), 4), "salu_share": round(cnt["salu"] / max(1, tot["salu"]), 4)})
    by_region.sort(key=lambda r: -(r.get("valu", 0) + r.get("lane", 0) + r.get("salu", 0)))
    out["by_region"] = by_region
    out["attribution"] = ("by_region: every instruction is charged to the TB_REGION point of the call site it was inlined into (the compiler's inline stack, kept by the assembly "
                          "listing behind every .loc); regions: the same, split by the innermost function the instruction came from (helpers like load_dom, mark_slice, lean_class_run_t)")
    out["regions"] = rows
    lines = []
    for (f, ln), cnt in sorted(per_line.items(), key=lambda kv: -(kv[1]["valu"] + kv[1]["lane"] + kv[1]["salu"]))[:60]:
        text = src[ln - 1].strip()[:140] if f == "kernels.hpp" and 0 < ln <= len(src) else ""
        lines.append({"file": f, "line": ln, **{k: round(per(cnt[k]), 1) for k in CLS if cnt[k]}, "source": text})
    out["hottest_source_lines"] = lines
    seg_rows.sort(key=lambda r: -r[0])
    out["hottest_segments"] = [{"segment": s["id"], "executions": round(per(c), 3), "instructions": s["counts"], "first_loc": s["first_loc"], "text": s["text"][:60]} for w, c, s in seg_rows[:40]]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        cmd_build(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "turbo_amd", "lib", "libturbo_hip_blocks.so"))
    elif sys.argv[1] == "table":
        cmd_table(*sys.argv[2:])
    else:
        print(__doc__)
