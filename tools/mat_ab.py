import os, sys, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "spart-python_amd"))
import torch
from spart_amd import workloads
from spart_amd.engine import Engine
B = 200_000
P = torch.as_tensor(workloads.lhs_params(B, "full").T.copy(), device="cuda:0")
allf = ("rdd", "rso", "rdo", "rsd", "leaf_refl", "leaf_tran", "soil_refl", "leaf_kchl", "soil_refl_dry")
for b in sys.argv[1:]:
    name, path = b.split("=")
    e = Engine("Sentinel2A-MSI", 0, lib_path=path)
    best = 1e9
    for r in range(5):
        o = e.run(P, "float32", materialize=allf); torch.cuda.synchronize(); del o
        e.profile(1); o = e.run(P, "float32", materialize=allf); torch.cuda.synchronize()
        ms, _ = e.profile_read(); e.profile(0); del o
        best = min(best, ms)
    print(f"{name}: 9 arrays band kernel {best:.3f} ms", flush=True)
