"""k_prelude time by part (leaf / soil / atmosphere / all), B = 200k: run under rocprofv3 --kernel-trace and read
the k_prelude rows in dispatch order (leaf, soil, atm, all; the canopy part = all - the rest)."""
import os, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
import torch
from spart_amd import workloads
from spart_amd.engine import Engine
B = 200_000
P = torch.as_tensor(workloads.lhs_params(B, "full").T.copy(), device="cuda:0")
e = Engine("Sentinel2A-MSI", 0)
for rep in range(3):
    e.prospect([P[i] for i in range(9)], "float32"); torch.cuda.synchronize()
    e.bsm([P[i] for i in range(9, 15)], "float32"); torch.cuda.synchronize()
    e.smac([P[i] for i in range(19, 22)], [P[i] for i in range(22, 26)]); torch.cuda.synchronize()
    e.run(P, "float32"); torch.cuda.synchronize()
