"""PCIe-inclusive LUT generation rate: host parameter table -> GPU -> host columns (no disk).

    python tools/lut_rate.py [B]      (prints the rate with and without the destination pre-faulting helper threads)"""
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
import numpy as np  # noqa: E402,F401
import torch  # noqa: E402
import spart_amd  # noqa: E402
from spart_amd import workloads  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
P = workloads.lhs_params(B, "full")
chunk = 1 << 20
spart_amd.generate_lut(P[:chunk], "Sentinel2A-MSI", chunk=chunk, prune=False)
torch.cuda.synchronize()
# prune=False: all 2162 bands of every spectrum evaluated (the function's default is the pruned column path)
for kw in (dict(fault_threads=0, prune=False), dict(fault_threads=4, prune=False), dict(fault_threads=8, prune=False),
           dict(fault_threads=12, prune=False), dict(fault_threads=8, prune=True), dict(fault_threads=0, prune=True)):
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        out = spart_amd.generate_lut(P, "Sentinel2A-MSI", chunk=chunk, **kw)
        best = min(best, time.perf_counter() - t0)
        del out
    print(f"B={B} chunk={chunk} {kw}: {B / best:.3e} spectra/s end to end (host table in, host columns out), {best * 1e3:.1f} ms", flush=True)
