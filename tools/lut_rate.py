"""PCIe-inclusive LUT generation rate: host parameter table -> GPU -> host columns (no disk)."""
import os, sys, time
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
import numpy as np, torch, spart_amd
from spart_amd import workloads
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
P = workloads.lhs_params(B, "full")
for chunk in (1 << 18, 1 << 20):
    spart_amd.generate_lut(P[:chunk], "Sentinel2A-MSI", chunk=chunk)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    spart_amd.generate_lut(P, "Sentinel2A-MSI", chunk=chunk)
    dt = time.perf_counter() - t0
    print(f"B={B} chunk={chunk}: {B/dt:.3e} spectra/s end to end (host table in, host columns out), {dt*1e3:.1f} ms")
    t0 = time.perf_counter()
    spart_amd.generate_lut(P, "Sentinel2A-MSI", chunk=chunk, prune=True)
    dt = time.perf_counter() - t0
    print(f"B={B} chunk={chunk} prune=True: {B/dt:.3e} spectra/s end to end, {dt*1e3:.1f} ms")
