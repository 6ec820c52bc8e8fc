import os, sys, time
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
import torch
from spart_amd import get_engine
eng = get_engine(None, 0)
for B, M in ((1_000_000, 4096), (1_000_000, 65536), (10_000_000, 4096)):
    lut = torch.rand((B, 13), device="cuda:0"); obs = torch.rand((M, 13), device="cuda:0")
    eng.lut_nearest(lut, obs); torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.lut_nearest(lut, obs); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"B={B} M={M}: {dt*1e3:.2f} ms, {B*M/dt:.3e} row-comparisons/s, LUT bytes x obs tiles / s = {B*13*4*((M+255)//256)/dt/1e9:.0f} GB/s (cached re-reads)")
