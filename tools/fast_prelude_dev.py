"""Deviation of the fast prelude (spart_materialize.fast_prelude: Newton LIDF + 8-point hot-spot panels) from the default
(literal) one, float64 columns, 1M rows of the config-4 and config-5 tables: distribution, the worst entries by the survey
metric |d| / max(|ref|, 1e-6) and by absolute deviation."""
import os, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
import numpy as np, torch
from spart_amd import get_engine, workloads
for kind, sensor in (("full", "Sentinel2A-MSI"), ("pro", "Sentinel2B-MSI")):
    eng = get_engine(sensor, 0)
    Ph = workloads.lhs_params(1_000_000, kind)
    P = torch.as_tensor(Ph.T.copy(), device="cuda:0")
    a = {k: v.clone() for k, v in eng.run(P, "float64", prune=True).items()}
    b = eng.run(P, "float64", prune=True, lidf="newton")
    for k in a:
        dabs = (b[k] - a[k]).abs()
        d = dabs / a[k].abs().clamp_min(1e-6)
        row = int(d.max(dim=1).values.argmax())
        q = torch.quantile(d.flatten()[::7].float(), torch.tensor([0.5, 0.999, 0.99999], device="cuda:0"))
        over = d > 1e-6
        print(f"{kind} {k}: max {float(d.max()):.2e} (row {row}: value {a[k][row].tolist()[int(d[row].argmax())]:.4e}), median {float(q[0]):.1e}, p99.9 {float(q[1]):.1e}, "
              f"p99.999 {float(q[2]):.1e}, entries > 1e-7: {int((d > 1e-7).sum())}, > 1e-6: {int(over.sum())} of {d.numel()}"
              f" (largest |value| among them {float(a[k].abs()[over].max()) if bool(over.any()) else 0:.2e})", flush=True)
        i = int(dabs.flatten().argmax()); r, c = divmod(i, dabs.shape[1])
        print(f"     largest ABSOLUTE deviation {float(dabs.max()):.2e} at row {r} band {c}: value {float(a[k][r, c]):.4e} (metric {float(d[r, c]):.2e}); "
              f"abs p99.999 {float(torch.quantile(dabs.flatten()[::7].float(), 0.99999)):.1e}", flush=True)
        print("     that row:", dict(zip(workloads.PARAM_NAMES, [float(f"{x:.4g}") for x in Ph[r]])))
    print("   worst row (metric) parameters:", dict(zip(workloads.PARAM_NAMES, [float(f"{x:.4g}") for x in Ph[row]])))
