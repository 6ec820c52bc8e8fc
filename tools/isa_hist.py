"""Instruction histogram of one kernel of libspart_hip (hipcc -S of the device code; no GPU needed).

    python tools/isa_hist.py k_prospectId [--loop]      (mangled-name fragment; --loop: only the hottest basic-block range)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
import build  # noqa: E402
frag = sys.argv[1]
with tempfile.TemporaryDirectory() as d:
    lines = [l for f in build.device_asm(d, sys.argv[2:]) for l in open(f).read().split("\n")]   # all translation units
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN5spart.*:", l) and frag in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = [l.strip() for l in lines[start + 1:end]]
ins = [l.split()[0] for l in body if l and not l.startswith((".", ";")) and not l.endswith(":")]
c = collections.Counter(ins)
valu = sum(v for k, v in c.items() if k.startswith("v_"))
print(lines[start].split(":")[0], "static instructions:", len(ins), "VALU:", valu, "f64 VALU:",
      sum(v for k, v in c.items() if k.startswith("v_") and "f64" in k))
for k, v in c.most_common(45):
    print(f"  {k:30s} {v}")
