import os, sys, time
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
import torch
from spart_amd import workloads
from spart_amd.engine import Engine
B = 200_000
P = torch.as_tensor(workloads.lhs_params(B, "full").T.copy(), device="cuda:0")
fields = ("leaf_refl", "leaf_tran", "leaf_kchl", "soil_refl", "soil_refl_dry", "rso", "rdo", "rsd", "rdd")
nbytes = (7 * 2162 + 2 * 2001) * 4 * B
engs = {b.split("=")[0]: Engine("Sentinel2A-MSI", 0, lib_path=b.split("=")[1]) for b in sys.argv[1:]}
for rnd in range(3):
    for n, e in engs.items():
        o = e.run(P, "float32", materialize=fields); torch.cuda.synchronize(); del o
        t0 = time.perf_counter(); o = e.run(P, "float32", materialize=fields); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        del o
        print(f"{n}: {dt*1e3:.3f} ms, {nbytes/dt/1e9:.0f} GB/s written", flush=True)
