"""Run ONE mode of the path for a few steps (the program rocprofv3 is pointed at, tools/collect_profiles.sh).

    python tools/mode_run.py materialized|pruned|lut_invert|headline [steps] [lib.so]
      materialized : spart_run_batch + the nine spectrum arrays, B = 200k, float32, padded row pitch (bench.py configs.materialized)
      pruned       : prune_unused_bands = 1, B = 1M, float32
      lut_invert   : spart_lut_nearest, 1M-row LUT x 65 536 observations, float32
      headline     : opt = NULL, B = 1M, float32"""
import os, sys, time
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
import torch
from spart_amd import workloads
from spart_amd.engine import Engine
mode = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
lib = sys.argv[3] if len(sys.argv) > 3 else None
FIELDS = ("leaf_refl", "leaf_tran", "leaf_kchl", "soil_refl", "soil_refl_dry", "rso", "rdo", "rsd", "rdd")
eng = Engine("Sentinel2A-MSI", 0, lib_path=lib)
if mode == "materialized":
    P = torch.as_tensor(workloads.lhs_params(200_000, "full").T.copy(), device="cuda:0")
    out = eng.run(P, "float32", materialize=FIELDS)
    fn = lambda: eng.run(P, "float32", materialize=FIELDS, out=out)
    unit, n = "spectra", 200_000
elif mode in ("pruned", "headline"):
    P = torch.as_tensor(workloads.lhs_params(1_000_000, "full").T.copy(), device="cuda:0")
    out = eng.run(P, "float32", prune=mode == "pruned")
    fn = lambda: eng.run(P, "float32", prune=mode == "pruned", out=out)
    unit, n = "spectra", 1_000_000
elif mode == "lut_invert":
    P = torch.as_tensor(workloads.lhs_params(1_000_000, "full").T.copy(), device="cuda:0")
    lut = eng.run(P, "float32", prune=True)["R_TOC"].clone()
    g = torch.Generator(device="cuda:0").manual_seed(3)
    obs = lut[torch.randint(0, lut.shape[0], (65536,), generator=g, device="cuda:0")] * (1 + 0.02 * torch.randn((65536, 13), generator=g, device="cuda:0"))
    fn = lambda: eng.lut_nearest(lut, obs)
    unit, n = "row comparisons", 1_000_000 * 65536
else:
    raise SystemExit(__doc__)
fn(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    fn()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"{mode}: {dt*1e3:.3f} ms per step, {n/dt:.4e} {unit}/s", flush=True)
