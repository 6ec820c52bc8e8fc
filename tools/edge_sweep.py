"""Randomised edge sweep (GPU box): fp64 and fp32 HIP columns against the oracle on parameters drawn from ranges wider
than the benchmark workloads, with edge values mixed in (LAI 0 / tiny / 10, dry soil, N = 1, zero pigments, hot spot
geometry, grazing angles, PRO leaves).  Test infrastructure (imports oracle/); prints a JSON summary."""
import json, multiprocessing as mp, os, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np

N = int(sys.argv[1]) if (__name__ == "__main__" and len(sys.argv) > 1) else 32768
SENSOR = "Sentinel2A-MSI"


def draw(n, seed=99):
    r = np.random.default_rng(seed)
    def u(lo, hi, edges=(), pe=0.15):
        x = r.uniform(lo, hi, n)
        if edges:
            m = r.random(n) < pe
            x[m] = r.choice(np.asarray(edges, dtype=float), m.sum())
        return x
    pro = r.random(n) < 0.3
    cols = [u(0, 100, (0, 1e-3)), np.where(pro, 0.0, u(0, 0.05, (0, 1e-5))), u(0, 0.1, (0, 1e-4)), u(0, 1.0, (0,)), u(0, 30, (0,)),
            u(0, 15, (0,)), u(1, 4, (1.0, 1.0001)), np.where(pro, u(0, 0.005, (0, 1e-6)), 0.0), np.where(pro, u(0, 0.02, (0, 1e-6)), 0.0),
            u(0.1, 1.0), u(-90, 90, (0, 90)), u(0, 180, (0, 100)), u(0, 80, (0, 5, 5.0001, 4.9)), u(5, 55, (25,)), u(0.001, 0.05, (0.015,)),
            u(0, 10, (0, 1e-4, 0.01, 10)), u(-1, 1), u(-1, 1), u(0.001, 0.5, (0.001, 0.5)),
            u(0, 80, (0, 30)), u(0, 70, (0, 30)), u(0, 360, (0, 180, 360, 90)),
            u(0.01, 1.0, (0.05,)), u(0.1, 0.6), u(0.1, 6.0), u(700, 1050, (1013.25,)), u(1, 366, (100,))]
    P = np.stack(cols, axis=1)
    s = np.abs(P[:, 16]) + np.abs(P[:, 17])            # |LIDFa| + |LIDFb| <= 1
    f = np.where(s > 1, 0.999 / s, 1.0)
    P[:, 16] *= f; P[:, 17] *= f
    hs = r.random(n) < 0.05                             # exact hot spot: tts == tto, psi == 0
    P[hs, 20] = P[hs, 19]; P[hs, 21] = 0.0
    return P


def worker(job):
    lo, hi = job
    import spart_oracle as O
    T = O.load_tables()
    P = draw(N)[lo:hi]
    out = {k: [] for k in ("R_TOC", "R_TOA", "L_TOA")}
    with np.errstate(all="ignore"):
        for i in range(0, len(P), 256):
            rr = O.spart_run(P[i:i + 256], SENSOR, T, pso="gl")
            for k in out:
                out[k].append(rr[k])
    return {k: np.concatenate(v) for k, v in out.items()}


def main():
    cores = min(16, len(os.sched_getaffinity(0)))
    with mp.get_context("fork").Pool(cores) as pool:
        parts = pool.map(worker, [(i * N // cores, (i + 1) * N // cores) for i in range(cores)])
    ref = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
    import torch
    from spart_amd import get_engine
    P = torch.as_tensor(draw(N).T.copy(), device="cuda:0")
    eng = get_engine(SENSOR, 0)
    res = {"rows": N}
    for dtype, floor, tol in (("float64", 1e-6, 1e-6), ("float32", 1e-3, 1e-4)):
        o = eng.run(P, dtype)
        for k in ref:
            x, r = o[k].double().cpu().numpy(), ref[k]
            fin = np.isfinite(r)
            rel = np.abs(x[fin] - r[fin]) / np.maximum(np.abs(r[fin]), floor)
            bad = np.argwhere(fin)[rel > tol]
            # rows whose leaves absorb in the short-wave infrared (some water, dry matter or protein / CBC): without any,
            # leaf reflectance + transmittance = 1 exactly there and the reference's canopy formulas divide by zero
            Ph = P.cpu().numpy().T
            absorbing = (Ph[:, 1] + Ph[:, 2] + Ph[:, 7] + Ph[:, 8]) > 1e-3
            relm = np.where(absorbing[:, None], np.abs(x - r) / np.maximum(np.abs(r), floor), 0.0)
            relm = np.where(np.isfinite(r), relm, 0.0)
            res[f"{dtype}/{k}"] = {"ref_finite": int(fin.sum()), "ref_nonfinite": int((~fin).sum()), "hip_nonfinite_where_ref_finite": int((~np.isfinite(x[fin])).sum()),
                                   "max_rel": float(np.nanmax(rel)), "over_tol": int((rel > tol).sum()), "first_bad_rows": [int(b[0]) for b in bad[:5]],
                                   "absorbing_rows": int(absorbing.sum()), "absorbing_max_rel": float(relm.max()), "absorbing_over_tol": int((relm > tol).sum()),
                                   "absorbing_bad_rows": [int(i) for i in np.unique(np.argwhere(relm > tol)[:, 0])[:8]]}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
