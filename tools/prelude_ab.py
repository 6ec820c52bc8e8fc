"""Per-stage times of the pruned step (prelude + column kernel) for several library builds.

    python tools/prelude_ab.py name=path.so ...   [--batch N]"""
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
import torch  # noqa: E402
from spart_amd import workloads  # noqa: E402
from spart_amd.engine import Engine  # noqa: E402

args = [a for a in sys.argv[1:] if "=" in a]
for B in (1_000_000, 100_000):
    P = torch.as_tensor(workloads.lhs_params(B, "full").T.copy(), device="cuda:0")
    for a in args:
        name, path = a.split("=", 1)
        e = Engine("Sentinel2A-MSI", 0, lib_path=path)
        for _ in range(3):
            e.run(P, "float32", prune=True)
        e.profile(10)
        for _ in range(10):
            e.run(P, "float32", prune=True)
        st, n = e.profile_read_stages()
        print(f"B={B:8d} {name:16s}", {k: round(v / n, 4) for k, v in st.items()}, flush=True)
