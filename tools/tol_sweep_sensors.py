import os, sys, json
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import spart_oracle as O
from spart_amd import workloads, get_engine, SENSORS
T = O.load_tables()
Ph = workloads.lhs_params(300_000, "full", seed=77)
P = torch.as_tensor(Ph.T.copy(), device="cuda:0")
for sensor in SENSORS:
    e = get_engine(sensor, 0)
    o64 = {k: v.clone() for k, v in e.run(P, "float64").items()}
    o32 = e.run(P, "float32")
    ref = O.spart_run(Ph[:512], sensor, T, pso="gl")
    row = {}
    for k in ("R_TOC", "R_TOA", "L_TOA"):
        d = (o32[k].double() - o64[k]).abs() / o64[k].abs().clamp_min(1e-6)
        g = o64[k][:512].cpu().numpy()
        row[k] = "f32-f64 max %.1e | f64-oracle max %.1e" % (float(d.max()), float(np.max(np.abs(g - ref[k]) / np.maximum(np.abs(ref[k]), 1e-6))))
    print(sensor, e.nb, json.dumps(row), flush=True)
