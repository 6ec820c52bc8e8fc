"""Where generate_lut's PCIe-inclusive time goes (8M spectra, pruned): fresh destination arrays against reused ones (page faults),
transparent-huge-page state of the box.    python tools/lut_host_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "spart-python_amd"))
import numpy as np, torch, spart_amd
from spart_amd import workloads
for f in ("enabled", "defrag"):
    try:
        print("THP", f, open(f"/sys/kernel/mm/transparent_hugepage/{f}").read().strip())
    except OSError as e:
        print("THP", f, e)
P = np.tile(workloads.lhs_params(1_000_000, "full"), (8, 1))
spart_amd.generate_lut(P[:1 << 18], "Sentinel2A-MSI")
for prune in (True, False):
    for label, kw in (("fresh destination", {}), ("fresh, no prefault threads", dict(fault_threads=0))):
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); o = spart_amd.generate_lut(P, "Sentinel2A-MSI", prune=prune, **kw); best = min(best, time.perf_counter() - t0)
        print(f"prune={prune} {label}: {best * 1e3:.1f} ms = {P.shape[0] / best:.3e} spectra/s", flush=True)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); spart_amd.generate_lut(P, "Sentinel2A-MSI", prune=prune, out=dict(o)); best = min(best, time.perf_counter() - t0)
    print(f"prune={prune} reused destination (out=): {best * 1e3:.1f} ms = {P.shape[0] / best:.3e} spectra/s", flush=True)
