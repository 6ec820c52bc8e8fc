import os, sys, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "spart-python_amd"))
import torch
from spart_amd import workloads
from spart_amd.engine import Engine
e = Engine("Sentinel2A-MSI", 0)
def t(f, n=5):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); o = f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0); del o
    return best * 1e3
for B, dt in ((1_000_000, "float32"), (1_000_000, "float64"), (10_000, "float64")):
    P = torch.as_tensor(workloads.lhs_params(B, "full").T.copy(), device="cuda:0")
    leaf = [P[i] for i in range(9)]; soil = [P[i] for i in range(9, 15)]
    es = 4 if dt == "float32" else 8
    ms = t(lambda: e.prospect(leaf, dt)); print(f"prospect B={B} {dt}: {ms:.3f} ms, {3*2001*es*B/ms/1e6:.0f} GB/s")
    ms = t(lambda: e.bsm(soil, dt)); print(f"bsm      B={B} {dt}: {ms:.3f} ms, {2*2001*es*B/ms/1e6:.0f} GB/s")
B = 300_000
P = torch.as_tensor(workloads.lhs_params(B, "full").T.copy(), device="cuda:0")
o = e.run(P, "float32", materialize=("leaf_refl", "leaf_tran", "soil_refl"))
can = [P[i] for i in range(15, 19)]; ang = [P[i] for i in range(19, 22)]
ms = t(lambda: e.sailh(o["leaf_refl"], o["leaf_tran"], o["soil_refl"], can, ang, "float32"))
print(f"sailh    B={B} float32: {ms:.3f} ms, {7*2162*4*B/ms/1e6:.0f} GB/s (3 read + 4 written)")
