"""The float32 step of spart_run_batch by batch size with its HIP-event stage times (profiles/r6_stage_probe.txt): what one GPU of an
N-way split of BASELINE config 4 does between gathers.

    python tools/stage_probe.py [B ...]        (SPART_CHUNK=<c> in the environment: samples per workgroup of the band kernel)"""
import sys, os, time
sys.path.insert(0, "spart-python_amd")
import torch
from spart_amd import get_engine, workloads
eng = get_engine("Sentinel2A-MSI", 0)
Pall = torch.as_tensor(workloads.lhs_params(1_000_000, "full").T.copy(), device="cuda:0")
for B in [int(x) for x in sys.argv[1:]] or [125000, 250000, 500000, 1000000]:
    P = Pall[:, :B].contiguous()
    out = {k: torch.empty((B, 13), dtype=torch.float32, device="cuda:0") for k in ("R_TOC", "R_TOA", "L_TOA")}
    for _ in range(20): eng.run(P, "float32", out=dict(out))
    torch.cuda.synchronize()
    n = 200 if B < 500000 else 40
    t0 = time.perf_counter()
    for _ in range(n): eng.run(P, "float32", out=dict(out))
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / n
    eng.profile(n)
    for _ in range(n): eng.run(P, "float32", out=dict(out))
    st, k = eng.profile_read_stages()
    eng.profile(0)
    print(f"CHUNK={os.environ.get('SPART_CHUNK','auto')} B={B}: step {sec*1e3:.4f} ms ({B/sec:.3e}/s; x{1e6/B:.0f} = {sec*1e3*1e6/B:.3f} ms per 1M)  stages " + " ".join(f"{a}={b/k:.4f}" for a, b in st.items()), flush=True)
