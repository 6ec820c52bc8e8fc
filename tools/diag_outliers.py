import os, sys, json
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import spart_oracle as O
from spart_amd import get_engine, workloads
kind, sensor = sys.argv[1], sys.argv[2]
B = 1_000_000
Ph = workloads.lhs_params(B, kind)
P = torch.as_tensor(Ph.T.copy(), device="cuda:0")
eng = get_engine(sensor, 0)
o64 = {k: v.clone() for k, v in eng.run(P, "float64").items()}
o32 = eng.run(P, "float32")
for k in ("R_TOC", "R_TOA"):
    e = ((o32[k].double() - o64[k]).abs() / o64[k].abs().clamp_min(float(os.environ.get("FLOOR", "1e-6"))))
    es, _ = e.max(dim=1)
    idx = torch.argsort(es, descending=True)[:6].cpu().numpy()
    print(k, "count >1e-4:", int((es > 1e-4).sum()), "count >1e-5:", int((es > 1e-5).sum()))
    for i in idx:
        j = int(e[i].argmax())
        print("  sample", i, "band", j, "err %.3e" % float(es[i]), "f64 %.6e f32 %.6e" % (float(o64[k][i, j]), float(o32[k][i, j])))
        print("    ", {n: round(float(v), 5) for n, v in zip(workloads.PARAM_NAMES, Ph[i])})
i = int(torch.argsort(((o32["R_TOA"].double() - o64["R_TOA"]).abs() / o64["R_TOA"].abs().clamp_min(1e-6)).max(dim=1).values, descending=True)[0])
ref = O.spart_run(Ph[i:i+1], sensor, pso="quad", full=True)
print("oracle R_TOA", ref["R_TOA"][0]); print("f64    R_TOA", o64["R_TOA"][i].cpu().numpy()); print("f32    R_TOA", o32["R_TOA"][i].cpu().numpy())
fields = ("rso", "rdo", "rsd", "rdd", "leaf_refl", "leaf_tran", "soil_refl")
Pi = P[:, i:i+1].contiguous()
m64 = {k: v.cpu().numpy() for k, v in eng.run(Pi, "float64", materialize=fields).items()}
m32 = {k: v.cpu().numpy() for k, v in eng.run(Pi, "float32", materialize=fields).items()}
for f in fields:
    d = np.abs(m32[f].astype(np.float64) - m64[f]) / np.maximum(np.abs(m64[f]), 1e-6)
    b = int(d.argmax()); print(f, "max rel %.3e at band %d: f64 %.6e f32 %.6e" % (d.max(), b, m64[f][0, b], m32[f][0, b]))
