"""Deviation of the fast prelude (spart_materialize.fast_prelude: Newton LIDF + 8-point hot-spot panels) from the default
(literal) one, float64 columns, 1M rows of the config-4 and config-5 tables: distribution and the worst row."""
import os, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
import numpy as np, torch
from spart_amd import get_engine, workloads
for kind, sensor in (("full", "Sentinel2A-MSI"), ("pro", "Sentinel2B-MSI")):
    eng = get_engine(sensor, 0)
    Ph = workloads.lhs_params(1_000_000, kind)
    P = torch.as_tensor(Ph.T.copy(), device="cuda:0")
    a = {k: v.clone() for k, v in eng.run(P, "float64").items()}
    b = eng.run(P, "float64", lidf="newton")
    for k in a:
        d = ((b[k] - a[k]).abs() / a[k].abs().clamp_min(1e-6))
        row = int(d.max(dim=1).values.argmax())
        q = torch.quantile(d.flatten()[::7].float(), torch.tensor([0.5, 0.999, 0.99999], device="cuda:0"))
        print(f"{kind} {k}: max {float(d.max()):.2e} (row {row}: value {a[k][row].tolist()[int(d[row].argmax())]:.4e}), median {float(q[0]):.1e}, p99.9 {float(q[1]):.1e}, "
              f"p99.999 {float(q[2]):.1e}, entries > 1e-7: {int((d > 1e-7).sum())}, > 1e-6: {int((d > 1e-6).sum())} of {d.numel()}", flush=True)
    print("   worst row parameters:", dict(zip(workloads.PARAM_NAMES, [float(f"{x:.4g}") for x in Ph[row]])))
