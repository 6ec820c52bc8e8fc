"""One LUT inversion workload for rocprofv3 (kernel times of centre / prep / scan / reduce / fallback / merge).

    rocprofv3 --kernel-trace --stats -d OUT -- python tools/lut_profile_run.py [uniform|correlated] [nb] [dtype]"""
import os, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
import torch
from spart_amd.engine import Engine
kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 13
dtype = sys.argv[3] if len(sys.argv) > 3 else "float32"
td = torch.float32 if dtype == "float32" else torch.float64
B, M = 1_000_000, 65536
g = torch.Generator("cuda:0").manual_seed(7)
eng = Engine(None, 0)
if kind == "uniform":
    lut = torch.rand((B, nb), device="cuda:0", dtype=td, generator=g); obs = torch.rand((M, nb), device="cuda:0", dtype=td, generator=g)
else:
    z = torch.rand((B, 4), device="cuda:0", dtype=torch.float64, generator=g)
    A = 1.5 * torch.randn((4, nb), device="cuda:0", dtype=torch.float64, generator=g)
    C = torch.randn((4, nb), device="cuda:0", dtype=torch.float64, generator=g)
    lut = (0.03 + 0.5 * torch.sigmoid(z @ A + (z * z) @ C - 1.0)).to(td)
    pick = torch.randint(0, B, (M,), device="cuda:0", generator=g)
    obs = (lut[pick].double() * (1 + 0.02 * torch.randn((M, nb), device="cuda:0", dtype=torch.float64, generator=g))).to(td)
for _ in range(6):
    _, _, st = eng.lut_nearest(lut, obs, dtype=dtype, stats=True)
torch.cuda.synchronize()
print(kind, nb, dtype, st)
