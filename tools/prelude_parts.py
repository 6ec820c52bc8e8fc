"""Which part of the per-sample prelude costs what: k_prelude is launched with one group of results at a time through the stage-level
entry points (leaf / soil / canopy / atmosphere masks), then with all of them (spart_run_batch, pruned), B = 200k, float64.
Run under `rocprofv3 --kernel-trace --output-format csv` and read the k_prelude dispatches in order (tools/prelude_parts.sh)."""
import os, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
import torch
from spart_amd import get_engine, workloads
B = 200_000
P = workloads.lhs_params(B, "full")
eng = get_engine("Sentinel2A-MSI", 0)
Pd = torch.as_tensor(P.T.copy(), device="cuda:0")
cols = [Pd[i] for i in range(27)]
for rep in range(3):
    r = eng.prospect(cols[0:9], "float32", outputs=("refl",))            # k_prelude: leaf group
    del r
    r = eng.bsm(cols[9:15], "float32")                                    # soil group
    del r
    z = torch.zeros((B, 2176), dtype=torch.float32, device="cuda:0")[:, :2162]
    r = eng.sailh(z + 0.1, z + 0.1, z + 0.2, cols[15:19], cols[19:22], "float32")   # canopy group
    del r, z
    r = eng.smac(cols[19:22], cols[22:26])                                # atmosphere group
    del r
    r = eng.run(Pd, "float64", prune=True)                                # all groups
    del r
    torch.cuda.synchronize()
print("done")
