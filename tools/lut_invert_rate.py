"""LUT inversion rate (spart_lut_nearest, whole call: centre + prep + scan + reduce + fallback + merge) on uniform random
LUTs and on a correlated one (4 latent parameters, 2 % noise), with the number of observations that took the brute-force path.

    python tools/lut_invert_rate.py [lib.so]"""
import os, sys, time
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
import torch
from spart_amd.engine import Engine
eng = Engine(None, 0, lib_path=sys.argv[1] if len(sys.argv) > 1 else None)
g = torch.Generator("cuda:0").manual_seed(7)


def correlated(B, M, nb, td):
    z = torch.rand((B, 4), device="cuda:0", dtype=torch.float64, generator=g)
    A = 1.5 * torch.randn((4, nb), device="cuda:0", dtype=torch.float64, generator=g)
    C = torch.randn((4, nb), device="cuda:0", dtype=torch.float64, generator=g)
    lut = (0.03 + 0.5 * torch.sigmoid(z @ A + (z * z) @ C - 1.0)).to(td)
    pick = torch.randint(0, B, (M,), device="cuda:0", generator=g)
    obs = (lut[pick].double() * (1 + 0.02 * torch.randn((M, nb), device="cuda:0", dtype=torch.float64, generator=g))).to(td)
    return lut, obs


for dtype, td in (("float32", torch.float32), ("float64", torch.float64)):
    for kind, B, M, nb in (("uniform", 1_000_000, 4096, 13), ("uniform", 1_000_000, 65536, 13), ("uniform", 10_000_000, 4096, 13),
                           ("uniform", 1_000_000, 65536, 21), ("uniform", 1_000_000, 65536, 6), ("correlated", 1_000_000, 65536, 13),
                           ("correlated", 1_000_000, 65536, 6), ("correlated", 1_000_000, 65536, 3), ("uniform", 1_000_000, 65536, 1)):
        if dtype == "float64" and B > 1_000_000:
            continue
        if kind == "uniform":
            lut = torch.rand((B, nb), device="cuda:0", dtype=td, generator=g); obs = torch.rand((M, nb), device="cuda:0", dtype=td, generator=g)
        else:
            lut, obs = correlated(B, M, nb, td)
        _, _, st = eng.lut_nearest(lut, obs, dtype=dtype, stats=True); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); eng.lut_nearest(lut, obs, dtype=dtype); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print(f"{dtype} {kind} B={B} M={M} nb={nb}: {best*1e3:.2f} ms, {B*M/best:.3e} row comparisons/s, {B*M*2*(nb+1)/best/1e12:.1f} Tflop/s "
              f"(2 (nb+1) flops each); brute-force path {st['brute_force']} of {M} observations, Nmax {st['nmax']:.3g}", flush=True)
