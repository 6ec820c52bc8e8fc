"""Count VALU instructions of a kernel's hottest loop whose three VGPR sources share a register bank (index mod 4): on gfx950
those issue at ~half rate (tools/ubench/bank_probe.hip).  hipcc -S of every translation unit (build.device_asm), no GPU.

    python tools/bank_conflicts.py <mangled-name fragment> [extra hipcc flags]"""
import os, re, sys, tempfile
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
import build  # noqa: E402


def regs(op):
    """VGPR indices an operand names: v5 -> [5]; v[4:5] -> [4] (the pair's bank is that of its first register); -v5, |v5|"""
    m = re.search(r"v\[(\d+):(\d+)\]", op)
    if m:
        return [int(m.group(1))]
    m = re.search(r"\bv(\d+)\b", op)
    return [int(m.group(1))] if m else []


def analyse(lines, frag):
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN5spart.*:", l) and frag in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end]
    # innermost loop = the "Depth=2" header .. the backward branch to it
    hdr = [i for i, l in enumerate(body) if "Inner Loop Header: Depth=2" in l]
    if hdr:
        lab = body[hdr[0] - 1].split(":")[0].strip()
        lo = hdr[0]
        hi = max(i for i, l in enumerate(body) if re.search(r"s_cbranch\w*\s+" + re.escape(lab) + r"\b|s_branch\s+" + re.escape(lab) + r"\b", l))
        body = body[lo:hi + 1]
    n3 = nv = same3 = same2 = 0
    bad = []
    for l in body:
        t = l.strip()
        if not t.startswith("v_") or t.startswith(("v_cmp", "v_mov", "v_cndmask", "v_readlane", "v_writelane")):
            continue
        nv += 1
        op, _, rest = t.partition(" ")
        ops = [o.strip() for o in rest.split(",")]
        srcs = ops[1:]
        if op.startswith(("v_fmac", "v_mac")):
            srcs = srcs + [ops[0]]                       # the destination is the addend
        r = [x for o in srcs for x in regs(o)]
        if len(r) >= 3:
            n3 += 1
            banks = [x % 4 for x in r[:3]]
            if banks[0] == banks[1] == banks[2]:
                same3 += 1
                bad.append(t)
            elif len(set(banks)) == 2:
                same2 += 1
    return nv, n3, same3, same2, bad


if __name__ == "__main__":
    frag = sys.argv[1]
    with tempfile.TemporaryDirectory() as d:
        lines = [l for f in build.device_asm(d, sys.argv[2:]) for l in open(f).read().split("\n")]
    nv, n3, same3, same2, bad = analyse(lines, frag)
    print(f"{frag}: {nv} VALU in the inner loop, {n3} with three VGPR sources, {same3} with all three in one bank, {same2} with two in one bank")
    for b in bad:
        print("   ", b)
