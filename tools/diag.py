"""Diagnosis of float32 / float64 / oracle deviations on the GPU box (one script, four sub-commands; the float32 runs
use f32_columns=True, i.e. the columns of the float32 band arithmetic itself -- the default float32 mode takes its
columns from a float64 pass and has nothing to diagnose):

    python tools/diag.py edges                         named edge rows (hot spot, tiny q, psi folding, PRO leaves ...)
    python tools/diag.py outliers <kind> <sensor>      1M LHS rows: the samples with the largest float32 deviation
    python tools/diag.py one <kind> <sensor> <row> <band>   one sample, every intermediate spectrum at one band
    python tools/diag.py sweep-rows <row> [<row> ...]  rows of tools/edge_sweep.py's draw, against the oracle
"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
for p in ("spart-python_amd", "oracle", "tools"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import spart_oracle as O  # noqa: E402
from spart_amd import get_engine, workloads  # noqa: E402

SPECTRA = ("rso", "rdo", "rsd", "rdd", "leaf_refl", "leaf_tran", "soil_refl", "soil_refl_dry")


def dev(P):
    return torch.as_tensor(P.T.copy(), device="cuda:0")


def f32(eng, P, **kw):
    return eng.run(P, "float32", f32_columns=True, **kw)


def edges():
    D = workloads.default_row
    names = ["hs30", "hs0", "q.001a", "q.001b", "q.5", "LAI.01", "LAI8", "psi270", "psi365", "psi-40", "graze", "SMp3", "SMp5",
             "N1", "N3", "PRO", "PRO0", "Cs1", "soilmax", "a-1", "a1", "b-1", "aot0", "gas0", "Pa500", "DOY1", "DOY365.5"]
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
        run = f32 if dtype == "float32" else (lambda e, p, **kw: e.run(p, "float64", **kw))
        out = run(eng, dev(P), materialize=("rso", "rdo", "rsd", "rdd"))
        for k, floor in (("rso", 1e-2), ("rdo", 1e-2), ("rsd", 1e-2), ("rdd", 1e-2), ("R_TOC", 1e-6), ("R_TOA", 1e-6)):
            e = np.abs(out[k].cpu().numpy().astype(np.float64) - ref[k]) / np.maximum(np.abs(ref[k]), floor)
            worst = np.argsort(e.max(axis=1))[::-1][:3]
            print(dtype, k, [(names[i], "%.1e" % e[i].max(), int(e[i].argmax())) for i in worst])


def outliers(kind, sensor):
    Ph = workloads.lhs_params(1_000_000, kind)
    P, eng = dev(Ph), get_engine(sensor, 0)
    o64 = {k: v.clone() for k, v in eng.run(P, "float64").items()}
    o32 = f32(eng, P)
    for k in ("R_TOC", "R_TOA"):
        e = (o32[k].double() - o64[k]).abs() / o64[k].abs().clamp_min(float(os.environ.get("FLOOR", "1e-6")))
        es, _ = e.max(dim=1)
        print(k, "samples >1e-4:", int((es > 1e-4).sum()), ">1e-5:", int((es > 1e-5).sum()))
        for i in torch.argsort(es, descending=True)[:6].cpu().numpy():
            j = int(e[i].argmax())
            print("  sample", i, "band", j, "err %.3e" % float(es[i]), "f64 %.6e f32 %.6e" % (float(o64[k][i, j]), float(o32[k][i, j])))
            print("    ", {n: round(float(v), 5) for n, v in zip(workloads.PARAM_NAMES, Ph[i])})


def one(kind, sensor, idx, band):
    Ph = workloads.lhs_params(1_000_000, kind)[idx:idx + 1]
    P, eng = dev(Ph), get_engine(sensor, 0)
    m64 = {k: v.cpu().numpy() for k, v in eng.run(P, "float64", materialize=SPECTRA).items()}
    m32 = {k: v.cpu().numpy() for k, v in f32(eng, P, materialize=SPECTRA).items()}
    ref = O.spart_run(Ph, sensor, pso="quad", full=True)
    wl = O.sensor_tables(O.load_tables(), sensor)["wl_smac"]
    b = int(wl[band]) - 400
    print("band centre", wl[band], "index", b)
    for f in SPECTRA:
        print(f, "f64 %.9e f32 %.9e rel %.2e" % (m64[f][0, b], m32[f][0, b], abs(m32[f][0, b] - m64[f][0, b]) / abs(m64[f][0, b])))
    for k in ("R_TOC", "R_TOA"):
        print(k, "oracle", ref[k][0, band], "f64", m64[k][0, band], "f32", m32[k][0, band])


def sweep_rows(rows):
    import edge_sweep as E
    P = E.draw(32768)
    with np.errstate(all="ignore"):
        ref = O.spart_run(P[rows], E.SENSOR, O.load_tables(), pso="gl", full=True)
    eng = get_engine(E.SENSOR, 0)
    fields = ("leaf_refl", "leaf_tran", "soil_refl", "rso", "rdo", "rsd", "rdd")
    o64, o32 = eng.run(dev(P[rows]), "float64", materialize=fields), f32(eng, dev(P[rows]))
    np.set_printoptions(precision=6, linewidth=200)
    rho, tau = O.pad_leaf(ref["leaf_refl"], ref["leaf_tran"])
    for i, r in enumerate(rows):
        print("row", r, {n: float("%.6g" % v) for n, v in zip(workloads.PARAM_NAMES, P[r])})
        print("  oracle R_TOC", ref["R_TOC"][i]); print("  hip64  R_TOC", o64["R_TOC"][i].cpu().numpy()); print("  hip32  R_TOC", o32["R_TOC"][i].cpu().numpy())
        for f, e in (("leaf_refl", rho), ("leaf_tran", tau), ("rso", ref["rso"]), ("rdd", ref["rdd"])):
            x, y = o64[f][i].cpu().numpy(), e[i]
            fin = np.isfinite(y)
            d = np.abs(x[fin] - y[fin]) / np.maximum(np.abs(y[fin]), 1e-9)
            if d.size:
                b = int(np.nanargmax(d))
                print("   ", f, "max rel %.2e at %d: hip %.6e ref %.6e; ref nonfinite %d" % (np.nanmax(d), b, x[fin][b], y[fin][b], (~fin).sum()))


if __name__ == "__main__":
    cmd, a = (sys.argv[1] if len(sys.argv) > 1 else "edges"), sys.argv[2:]
    if cmd == "edges":
        edges()
    elif cmd == "outliers":
        outliers(a[0], a[1])
    elif cmd == "one":
        one(a[0], a[1], int(a[2]), int(a[3]))
    elif cmd == "sweep-rows":
        sweep_rows([int(x) for x in a] or [10012, 1410, 1602, 1818, 2084, 136])
    else:
        raise SystemExit(__doc__)
