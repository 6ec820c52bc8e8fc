import os, sys, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "spart-python_amd"))
import torch
from spart_amd import workloads
from spart_amd.engine import Engine
B = 200_000
P = torch.as_tensor(workloads.lhs_params(B, "full").T.copy(), device="cuda:0")
allf = ("rdd", "rso", "rdo", "rsd", "leaf_refl", "leaf_tran", "soil_refl", "leaf_kchl", "soil_refl_dry")
e = Engine("Sentinel2A-MSI", 0)
for n in (0, 1, 2, 4, 7, 9):
    fields = allf[:n]
    best = 1e9
    for r in range(4):
        o = e.run(P, "float32", materialize=fields); torch.cuda.synchronize(); del o
        e.profile(1)
        o = e.run(P, "float32", materialize=fields); torch.cuda.synchronize()
        ms, _ = e.profile_read(); e.profile(0); del o
        best = min(best, ms)
    nbytes = sum((2162 if f not in ("leaf_kchl", "soil_refl_dry") else 2001) for f in fields) * 4 * B
    print(f"{n} arrays: band kernel {best:.3f} ms, {nbytes/best/1e6:.0f} GB/s", flush=True)
