import os, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import spart_oracle as O
from spart_amd import get_engine, workloads
D = workloads.default_row
names = ["hs30", "hs0", "q.001a", "q.001b", "q.5", "LAI.01", "LAI8", "psi270", "psi365", "psi-40", "graze", "SMp3", "SMp5", "N1", "N3", "PRO", "PRO0", "Cs1", "soilmax", "a-1", "a1", "b-1", "aot0", "gas0", "Pa500", "DOY1", "DOY365.5"]
rows = [D(tts=30, tto=30, psi=0), D(tts=0, tto=0, psi=0), D(q=0.001, tts=60, tto=30, psi=160),
        D(q=0.001, tts=5, tto=5, psi=1), D(q=0.5), D(LAI=0.01), D(LAI=8), D(psi=270), D(psi=365), D(psi=-40),
        D(tts=80, tto=60, psi=90), D(SMp=3), D(SMp=5), D(N=1.0), D(N=3.0, Cab=80, Cw=0.05),
        D(PROT=0.003, CBC=0.01), D(Cdm=0.0, PROT=0.001, CBC=0.0), D(Cs=1.0), D(B=0.9, lat=30, lon=120, SMp=55),
        D(LIDFa=-1, LIDFb=0), D(LIDFa=1, LIDFb=0), D(LIDFa=0, LIDFb=-1), D(aot550=0.0), D(uh2o=0.0, uo3=0.0),
        D(Pa=500.0), D(DOY=1), D(DOY=365.5)]
P = np.concatenate(rows)
ref = O.spart_run(P, "Sentinel2A-MSI", pso="quad", full=True)
eng = get_engine("Sentinel2A-MSI", 0)
for dtype in ("float32", "float64"):
    out = eng.run(torch.as_tensor(P.T.copy(), device="cuda:0"), dtype, materialize=("rso", "rdo", "rsd", "rdd"))
    for k in ("rso", "rdo", "rsd", "rdd"):
        g = out[k].cpu().numpy().astype(np.float64)
        e = np.abs(g - ref[k]) / np.maximum(np.abs(ref[k]), 1e-2)
        worst = np.argsort(e.max(axis=1))[::-1][:3]
        print(dtype, k, [(names[i], "%.1e" % e[i].max(), int(e[i].argmax())) for i in worst])
    for k in ("R_TOC", "R_TOA"):
        g = out[k].cpu().numpy().astype(np.float64)
        e = np.abs(g - ref[k]) / np.maximum(np.abs(ref[k]), 1e-3)
        worst = np.argsort(e.max(axis=1))[::-1][:3]
        print(dtype, k, [(names[i], "%.1e" % e[i].max()) for i in worst])
a = ref["aux"]
i = names.index("q.001b")
print("q.001b: k K dso sumPso Pso2w", a["k"][i], a["K"][i], a["dso"][i], a["sumPso"][i], a["Pso2w"][i])
