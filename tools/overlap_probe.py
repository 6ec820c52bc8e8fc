"""Probe: consecutive steps issued on two alternating streams (own workspace and result buffers each), so that the
prelude / slot pass / sensor kernel of step i+1 overlap the full-band kernel of step i -- against the same steps on one
stream.

    python tools/overlap_probe.py [B]"""
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
import torch  # noqa: E402
from spart_amd import get_engine, workloads  # noqa: E402

for B in ([int(sys.argv[1])] if len(sys.argv) > 1 else [1_000_000, 125_000]):
    eng = get_engine("Sentinel2A-MSI", 0)
    P = torch.as_tensor(workloads.lhs_params(B, "full").T.copy(), device="cuda:0")
    n = int(eng.lib.spart_workspace_bytes(eng.ctx, 0, B))
    for nstream in (1, 2, 3):
        streams = [torch.cuda.Stream() for _ in range(nstream)]
        ws = [torch.empty(n, dtype=torch.uint8, device="cuda:0") for _ in range(nstream)]
        outs = [{k: torch.empty((B, eng.nb), dtype=torch.float32, device="cuda:0") for k in ("R_TOC", "R_TOA", "L_TOA")} for _ in range(nstream)]
        torch.cuda.synchronize()

        def run(steps):
            for i in range(steps):
                j = i % nstream
                with torch.cuda.stream(streams[j]):
                    eng.run(P, "float32", out=outs[j], _workspace=ws[j])
            torch.cuda.synchronize()
        run(4)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            run(20)
            best = min(best, (time.perf_counter() - t0) / 20)
        ref = eng.run(P, "float32")
        ok = all(torch.equal(ref[k], outs[j][k]) for j in range(nstream) for k in ref)
        print(f"B={B} streams={nstream}: {best * 1e3:.3f} ms/step  {B / best:.3e} spectra/s  identical={ok}", flush=True)
