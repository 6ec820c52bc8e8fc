"""Materialised-spectra rate (the HBM-store-bound mode of spart_run_batch): dense rows vs padded row pitch.

    python tools/mat_bench.py [lib.so]
"""
import os, sys, time
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
import torch
from spart_amd import workloads
from spart_amd.engine import Engine, ROW_PITCH
B = 200_000
P = torch.as_tensor(workloads.lhs_params(B, "full").T.copy(), device="cuda:0")
fields = ("leaf_refl", "leaf_tran", "leaf_kchl", "soil_refl", "soil_refl_dry", "rso", "rdo", "rsd", "rdd")
nbytes = (7 * 2162 + 2 * 2001) * 4 * B
lib = sys.argv[1] if len(sys.argv) > 1 else None
engs = {"dense rows (2162/2001)": Engine("Sentinel2A-MSI", 0, lib_path=lib, row_pitch=None),
        "padded rows %s" % (ROW_PITCH,): Engine("Sentinel2A-MSI", 0, lib_path=lib)}
ref = None
for rnd in range(3):
    for n, e in engs.items():
        o = e.run(P, "float32", materialize=fields); torch.cuda.synchronize()
        if rnd == 0:
            cur = {k: o[k].contiguous() for k in fields}
            if ref is None:
                ref = cur
            else:
                assert all(torch.equal(ref[k], cur[k]) for k in fields), "padded and dense spectra differ"
            del cur
        del o
        e.profile(1)
        t0 = time.perf_counter(); o = e.run(P, "float32", materialize=fields); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        ms, _ = e.profile_read(); e.profile(0)
        del o
        print(f"{n}: step {dt*1e3:.3f} ms (band kernel {ms:.3f} ms), {B/dt:.3e} spectra/s, {nbytes/dt/1e9:.0f} GB/s of spectra written "
              f"({nbytes/ms/1e6:.0f} GB/s inside the band kernel)", flush=True)
