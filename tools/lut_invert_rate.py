"""LUT inversion rate (spart_lut_nearest, whole call: prep + scan + reduce).

    python tools/lut_invert_rate.py [lib.so]"""
import os, sys, time
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
import torch
from spart_amd.engine import Engine
eng = Engine(None, 0, lib_path=sys.argv[1] if len(sys.argv) > 1 else None)
for dtype, td in (("float32", torch.float32), ("float64", torch.float64)):
    for B, M, nb in ((1_000_000, 4096, 13), (1_000_000, 65536, 13), (10_000_000, 4096, 13), (1_000_000, 65536, 21), (1_000_000, 65536, 6)):
        if dtype == "float64" and B > 1_000_000:
            continue
        lut = torch.rand((B, nb), device="cuda:0", dtype=td); obs = torch.rand((M, nb), device="cuda:0", dtype=td)
        eng.lut_nearest(lut, obs, dtype=dtype); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); eng.lut_nearest(lut, obs, dtype=dtype); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print(f"{dtype} B={B} M={M} nb={nb}: {best*1e3:.2f} ms, {B*M/best:.3e} row comparisons/s, {B*M*2*(nb+1)/best/1e12:.1f} Tflop/s (2 (nb+1) flops each)", flush=True)
