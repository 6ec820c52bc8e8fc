import os, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import spart_oracle as O
from spart_amd import get_engine, workloads
kind, sensor, idx, band = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
Ph = workloads.lhs_params(1_000_000, kind)[idx:idx+1]
eng = get_engine(sensor, 0)
P = torch.as_tensor(Ph.T.copy(), device="cuda:0")
fields = ("rso", "rdo", "rsd", "rdd", "leaf_refl", "leaf_tran", "soil_refl", "soil_refl_dry")
m64 = {k: v.cpu().numpy() for k, v in eng.run(P, "float64", materialize=fields).items()}
m32 = {k: v.cpu().numpy() for k, v in eng.run(P, "float32", materialize=fields).items()}
ref = O.spart_run(Ph, sensor, pso="quad", full=True)
wl = O.sensor_tables(O.load_tables(), sensor)["wl_smac"]
b = int(wl[band]) - 400
print("band centre", wl[band], "index", b)
for f in fields:
    print(f, "f64 %.9e f32 %.9e rel %.2e" % (m64[f][0, b], m32[f][0, b], abs(m32[f][0, b] - m64[f][0, b]) / abs(m64[f][0, b])))
for k in ("R_TOC", "R_TOA"):
    print(k, "oracle", ref[k][0, band], "f64", m64[k][0, band], "f32", m32[k][0, band])
for k in ("atm_Ta_ss", "atm_Ta_sd", "atm_Ta_oo", "atm_Ta_do", "atm_Ra_dd", "atm_Ra_so", "atm_Tg"):
    print(k, ref[k][0, band])
