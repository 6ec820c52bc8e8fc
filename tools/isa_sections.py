"""Instruction budget of the full-band kernel's SAMPLE LOOP by model section, from the ISA (no GPU needed).

    python tools/isa_sections.py [--md]

Each translation unit is compiled to assembly with its own flags (build.TU_FLAGS) plus -gline-tables-only, which adds `.loc`
directives carrying the whole inline chain of every instruction and does not change the code (the instruction count equals the
product build's; checked below).  The sample loop of k_bands is found as the innermost loop (backward branch) with the most
VALU instructions; every instruction in it is attributed to the line of k_bands (spart_kernels.h) at the OUTER end of its inline
chain, and that line to a section of the model by what the source line calls:

    leaf_band     PROSPECT-5D / PRO plate model             (prospect_5d.py:170-241)
    soil          soil_dry + soil_band: BSM + soilwat        (bsm.py:49-52, 99-124)
    canopy_core   SAILH without the soil background          (sailh.py:142-214)
    canopy_soil   coupling with the soil + the 4 outputs      (sailh.py:216-233)
    staging       per-sample constants LDS -> VGPR, the double-buffered copy of the next 32 samples
    select/sum    thermal-band selects, band-sum accumulation
    loop          loop control, address arithmetic, compiler-generated moves without a source line

Counts are STATIC instructions of the loop body, split into the straight-line part every (wave, sample) issues and the part
inside forward-skipped regions (`s_cbranch_execz / vccz` over a block: the regime branches of the plate model's tau(K) and
ln(1 + x) forms, the singular-J branch of SAILH, the last tile's thermal path), which a wave issues only when one of its
lanes needs it ("cond").  The dynamic count per (wave, sample) -- SQ_INSTS_VALU / (32 waves x samples) from the committed
counter passes: 272 float32, 376 float64 -- lies between `always` and `always + cond`."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
import build  # noqa: E402

KH = os.path.join(build.CSRC, "spart_kernels.h")
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")
SECTIONS = ["leaf_band", "soil", "canopy_core", "canopy_soil", "staging", "select/sum", "loop"]


def section_of_source_lines():
    """line number of spart_kernels.h (inside k_bands) -> section"""
    src = open(KH).read().split("\n")
    start = next(i for i, l in enumerate(src) if l.startswith("void k_bands("))
    end = next(i for i in range(start, len(src)) if src[i].startswith("// K3:"))
    out = {}
    for i in range(start, end):
        l = src[i]
        if "leaf_band<" in l:
            s = "leaf_band"
        elif "canopy_core<" in l or "load_canopy<" in l:
            s = "canopy_core"
        elif "canopy_soil<" in l:
            s = "canopy_soil"
        elif any(k in l for k in ("soil_dry<", "soil_band", "soil_tw1", "T fm[7]", "film_same", "C_FILM2L")):
            s = "soil"
        elif any(k in l for k in ("stage_fetch", "stage_put", "stage_constants", "lc[i] =", "lds_c", "lds_all")):
            s = "staging"
        elif "sum_" in l or "thermal ?" in l:
            s = "select/sum"
        else:
            s = "loop"
        out[i + 1] = s
    return out, start + 1, end


def asm_with_lines(src):
    with tempfile.TemporaryDirectory() as d:
        o = os.path.join(d, "k.s")
        subprocess.check_call([build.hipcc(), *build.tu_flags(src), "-DSPART_FAST_MATH=1", "-gline-tables-only", "-S", "--cuda-device-only",
                               "-o", o, src], stderr=subprocess.DEVNULL)
        return open(o).read().split("\n")


def kernel_body(lines, frag):
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN5spart.*:", l) and frag in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    return lines[start].split(":")[0], lines[start + 1:end]


def sample_loop(body):
    """(first, last) index into body of the innermost loop with the most VALU instructions"""
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    nv = lambda a, b: sum(1 for l in body[a:b + 1] if l.strip().startswith("v_"))          # noqa: E731
    inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] and nv(*o) > 0.5 * nv(*lp) for o in loops)]
    return max(inner, key=lambda lp: nv(*lp))


def budget(lines, frag, secmap, k0, k1):
    name, body = kernel_body(lines, frag)
    a, b = sample_loop(body)
    cur = "loop"
    counts = {s: collections.Counter() for s in SECTIONS}
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    skip_until = []                                  # body indices of the labels of forward skips we are inside of
    for i in range(a, b + 1):
        l = body[i]
        t = l.strip()
        while skip_until and i >= skip_until[-1]:
            skip_until.pop()
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", t)
        if m and i < labels.get(m.group(1), -1) <= b:
            skip_until.append(labels[m.group(1)])
            skip_until.sort(reverse=True)
        if t.startswith(".loc"):
            # outermost frame of the inline chain that lies inside k_bands
            frames = re.findall(r"(\S+?):(\d+):\d+", t.split(";", 1)[1]) if ";" in t else []
            cur = "loop"
            for f, ln in reversed(frames):
                if f.endswith("spart_kernels.h") and k0 <= int(ln) <= k1:
                    cur = secmap.get(int(ln), "loop")
                    break
            continue
        if not t or t.startswith((".", ";")) or t.endswith(":"):
            continue
        op = t.split()[0]
        kind = ("trans" if op.startswith(TRANS) else "valu") if op.startswith("v_") else ("lds" if op.startswith("ds_") else
                ("vmem" if op.startswith(("global_", "buffer_", "flat_")) else ("salu" if op.startswith("s_") else "other")))
        counts[cur][kind] += 1
        if skip_until and kind in ("valu", "trans"):
            counts[cur]["cond"] += 1
    return name, counts, sum(1 for l in body if l.strip().startswith("v_"))


def product_valu_count(src, frag):
    with tempfile.TemporaryDirectory() as d:
        o = os.path.join(d, "k.s")
        subprocess.check_call([build.hipcc(), *build.tu_flags(src), "-DSPART_FAST_MATH=1", "-S", "--cuda-device-only", "-o", o, src],
                              stderr=subprocess.DEVNULL)
        _, body = kernel_body(open(o).read().split("\n"), frag)
    return sum(1 for l in body if l.strip().startswith("v_"))


def main():
    secmap, k0, k1 = section_of_source_lines()
    rows = []
    for src, frag, label in ((build.SOURCES[1], "k_bandsIfLi0ELi1ELb0E", "float32 `k_bands<float,0,1,false>`"),
                             (build.SOURCES[0], "k_bandsIdLi0ELi1ELb0E", "float64 `k_bands<double,0,1,false>`")):
        lines = asm_with_lines(src)
        name, counts, total_g = budget(lines, frag, secmap, k0, k1)
        assert total_g == product_valu_count(src, frag), "-gline-tables-only changed the code"
        rows.append((label, counts))
    md = "--md" in sys.argv
    if md:
        print("| section | " + " | ".join(f"{lab}: VALU always + cond (transcendental) / LDS / SALU" for lab, _ in rows) + " |")
        print("|---|" + "---|" * len(rows))
    for s in SECTIONS + ["TOTAL"]:
        cells = []
        for _, counts in rows:
            cs = [counts[s]] if s != "TOTAL" else list(counts.values())
            v = sum(c["valu"] + c["trans"] for c in cs)
            t = sum(c["trans"] for c in cs)
            cd = sum(c["cond"] for c in cs)
            cells.append(f"{v - cd} + {cd} ({t}) / {sum(c['lds'] for c in cs)} / {sum(c['salu'] for c in cs)}")
        print(("| " + s + " | " + " | ".join(cells) + " |") if md else f"{s:12s} " + "   ".join(f"{c:>22s}" for c in cells))


if __name__ == "__main__":
    main()
