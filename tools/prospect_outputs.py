import sys, time
sys.path.insert(0, "/root/repo/spart-python_amd")
import torch
from spart_amd import get_engine, workloads
eng = get_engine(None, 0)
for B in (1_000_000, 10_000):
    P = workloads.lhs_params(B, "leaf")
    cols = [torch.as_tensor(P[:, i].copy(), device="cuda:0") for i in range(9)]
    for dtype in ("float64", "float32"):
        for outs in ((), ("refl",), ("refl", "tran"), ("refl", "tran", "kChlrel")):
            reps = 5 if B > 100000 else 100
            for _ in range(3): eng.prospect(cols, dtype, outputs=outs)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(reps): eng.prospect(cols, dtype, outputs=outs)
            torch.cuda.synchronize(); sec = (time.perf_counter() - t0) / reps
            print(f"B={B} {dtype} outputs={len(outs)}: {sec*1e3:.4f} ms", flush=True)
