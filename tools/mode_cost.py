"""Step time of spart_run_batch per mode on one GPU: float32 default (float64 column kernel), float32 legacy columns
(f32_columns), float64; B = 1M and 100k.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
import torch  # noqa: E402
from spart_amd import get_engine, workloads  # noqa: E402

sensor = sys.argv[1] if len(sys.argv) > 1 else "Sentinel2A-MSI"
eng = get_engine(sensor, 0)
for B in (1_000_000, 125_000, 100_000):
    P = torch.as_tensor(workloads.lhs_params(B, "full").T.copy(), device="cuda:0")
    for name, dtype, kw in (("f32 default", "float32", {}), ("f32 legacy columns", "float32", dict(f32_columns=True)),
                            ("f64", "float64", {}), ("f32 pruned", "float32", dict(prune=True))):
        out = {k: torch.empty((B, eng.nb), dtype=torch.float32 if dtype == "float32" else torch.float64, device="cuda:0")
               for k in ("R_TOC", "R_TOA", "L_TOA")}
        for _ in range(3):
            eng.run(P, dtype, out=dict(out), **kw)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(10):
                eng.run(P, dtype, out=dict(out), **kw)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 10)
        print(f"B={B:8d} {name:20s} {best * 1e3:8.3f} ms/step  {B / best:.3e} spectra/s", flush=True)
