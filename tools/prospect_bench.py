"""BASELINE config 2 under the profiler: PROSPECT-5D leaf only (spart_prospect_batch), 10k LeafBiology samples x 2001
bands, float64, refl + tran + kChlrel out (48 096 B per leaf spectrum).  Prints the synchronised call time; run under
rocprofv3 (--stats / --pmc) for the kernel-level numbers (tools/collect_profiles.sh).

    python tools/prospect_bench.py [B] [dtype] [reps]"""
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
import torch  # noqa: E402
from spart_amd import get_engine, workloads  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
dtype = sys.argv[2] if len(sys.argv) > 2 else "float64"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
eng = get_engine(None, 0)
P = workloads.lhs_params(B, "leaf")
cols = [torch.as_tensor(P[:, i].copy(), device="cuda:0") for i in range(9)]
for _ in range(5):
    out = eng.prospect(cols, dtype)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    out = eng.prospect(cols, dtype)
torch.cuda.synchronize()
sec = (time.perf_counter() - t0) / reps
es = 8 if dtype == "float64" else 4
by = 9 * 8 + 3 * 2001 * es
print(f"prospect B={B} {dtype}: {sec * 1e3:.4f} ms/call  {B / sec:.3e} leaf spectra/s  {by * B / sec / 1e9:.1f} GB/s algorithmic "
      f"({by * B / sec / 8e12:.3f} of 8 TB/s)", flush=True)
