"""Probe: does splitting the batch over two streams (so that the fp64 prelude / sensor kernels of one half run
beside the band kernel of the other half) shorten a step?"""
import os, sys, time, statistics
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
import torch
from spart_amd import workloads
from spart_amd.engine import Engine
B = 1_000_000
P = torch.as_tensor(workloads.lhs_params(B, "full").T.copy(), device="cuda:0")
e1, e2 = Engine("Sentinel2A-MSI", 0), Engine("Sentinel2A-MSI", 0)
res = torch.empty((3, B, 13), device="cuda:0")
def out(lo, hi): return {"R_TOC": res[0, lo:hi], "R_TOA": res[1, lo:hi], "L_TOA": res[2, lo:hi]}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def single():
    e1.run(P, "float32", out=out(0, B))
def split(nparts):
    bounds = [B * i // nparts for i in range(nparts + 1)]
    for i in range(nparts):
        st, eng = (s1, e1) if i % 2 == 0 else (s2, e2)
        with torch.cuda.stream(st):
            lo, hi = bounds[i], bounds[i + 1]
            eng.run(P[:, lo:hi].contiguous() if False else P[:, lo:hi], "float32", out=out(lo, hi))
Ph = [P[:, :B // 2].contiguous(), P[:, B // 2:].contiguous()]
def split2():
    with torch.cuda.stream(s1):
        e1.run(Ph[0], "float32", out=out(0, B // 2))
    with torch.cuda.stream(s2):
        e2.run(Ph[1], "float32", out=out(B // 2, B))
Pq = [P[:, B * i // 4: B * (i + 1) // 4].contiguous() for i in range(4)]
def split4():
    for i in range(4):
        st, eng = (s1, e1) if i % 2 == 0 else (s2, e2)
        with torch.cuda.stream(st):
            eng.run(Pq[i], "float32", out=out(B * i // 4, B * (i + 1) // 4))
for name, fn in (("single", single), ("split2", split2), ("split4", split4), ("single", single), ("split2", split2)):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(8):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{name:8s} min {min(ts):.3f} med {statistics.median(ts):.3f} ms")
