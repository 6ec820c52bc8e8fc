"""Interleaved A/B of libspart_hip builds on the materialised mode (9 spectrum arrays, B = 200k, float32, padded pitch):
band-kernel HIP-event time and whole-step wall time per build, rounds interleaved in one process.

    python tools/mat_ab.py name1=path1.so name2=path2.so [--chunk N]"""
import os, sys, time, statistics
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "spart-python_amd"))
import torch
from spart_amd import workloads
from spart_amd.engine import Engine
B = 200_000
P = torch.as_tensor(workloads.lhs_params(B, "full").T.copy(), device="cuda:0")
allf = ("rdd", "rso", "rdo", "rsd", "leaf_refl", "leaf_tran", "soil_refl", "leaf_kchl", "soil_refl_dry")
nbytes = (7 * 2162 + 2 * 2001) * 4 * B
engs, outs = {}, {}
for b in [a for a in sys.argv[1:] if "=" in a]:
    name, path = b.split("=")
    engs[name] = Engine("Sentinel2A-MSI", 0, lib_path=path)
    outs[name] = engs[name].run(P, "float32", materialize=allf)
torch.cuda.synchronize()
first = next(iter(outs))
for n in outs:
    assert all(torch.equal(outs[n][k], outs[first][k]) for k in allf), n
band = {n: [] for n in engs}; wall = {n: [] for n in engs}
for r in range(7):
    for n, e in engs.items():
        e.profile(1); torch.cuda.synchronize(); t0 = time.perf_counter()
        e.run(P, "float32", materialize=allf, out=outs[n]); torch.cuda.synchronize(); wall[n].append((time.perf_counter() - t0) * 1e3)
        ms, _ = e.profile_read(); e.profile(0); band[n].append(ms)
for n in engs:
    print(f"{n:16s} band min {min(band[n]):.3f} med {statistics.median(band[n]):.3f} ms = {nbytes/min(band[n])/1e6:.0f} GB/s | step min {min(wall[n]):.3f} med "
          f"{statistics.median(wall[n]):.3f} ms = {B/min(wall[n])*1e3:.3e} spectra/s", flush=True)
