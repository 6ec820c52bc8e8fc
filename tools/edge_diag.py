import os, sys, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import spart_oracle as O
import edge_sweep as E
from spart_amd import get_engine, workloads
P = E.draw(32768)
rows = [10012, 1410, 1602, 1818, 2084, 136]
T = O.load_tables()
eng = get_engine(E.SENSOR, 0)
with np.errstate(all="ignore"):
    ref = O.spart_run(P[rows], E.SENSOR, T, pso="gl", full=True)
Pd = torch.as_tensor(P[rows].T.copy(), device="cuda:0")
fields = ("leaf_refl", "leaf_tran", "soil_refl", "rso", "rdo", "rsd", "rdd")
o64 = eng.run(Pd, "float64", materialize=fields); o32 = eng.run(Pd, "float32")
np.set_printoptions(precision=6, linewidth=200)
for i, r in enumerate(rows):
    print("row", r, {n: float("%.6g" % v) for n, v in zip(workloads.PARAM_NAMES, P[r])})
    print("  oracle R_TOC", ref["R_TOC"][i]); print("  hip64  R_TOC", o64["R_TOC"][i].cpu().numpy()); print("  hip32  R_TOC", o32["R_TOC"][i].cpu().numpy())
    rho, tau = O.pad_leaf(ref["leaf_refl"], ref["leaf_tran"])
    for f, e in (("leaf_refl", rho), ("leaf_tran", tau), ("rso", ref["rso"]), ("rdd", ref["rdd"])):
        x = o64[f][i].cpu().numpy(); y = e[i]
        fin = np.isfinite(y)
        d = np.abs(x[fin] - y[fin]) / np.maximum(np.abs(y[fin]), 1e-9)
        b = int(np.nanargmax(d)) if d.size else -1
        print("   ", f, "max rel %.2e at %d: hip %.6e ref %.6e; ref nonfinite %d" % (np.nanmax(d) if d.size else 0, b, x[fin][b] if d.size else 0, y[fin][b] if d.size else 0, (~fin).sum()))
