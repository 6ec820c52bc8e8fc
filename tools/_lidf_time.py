import os, sys, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "spart-python_amd"))
import torch
from spart_amd import get_engine, workloads
eng = get_engine("Sentinel2A-MSI", 0)
P = torch.as_tensor(workloads.lhs_params(1_000_000, "full").T.copy(), device="cuda:0")
a, b = P[16].contiguous(), P[17].contiguous()
for _ in range(3): eng.lidf(a, b)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): eng.lidf(a, b)
torch.cuda.synchronize(); print("k_lidf 1M:", (time.perf_counter() - t0) / 10 * 1e3, "ms")
for _ in range(3): eng.run(P, "float32", prune=True)
eng.profile(10)
for _ in range(10): eng.run(P, "float32", prune=True)
st, n = eng.profile_read_stages(); eng.profile(0)
print({k: v / n for k, v in st.items()})
