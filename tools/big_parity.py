"""Parity at scale on the GPU box: HIP fp64 / fp32 sensor columns against the oracle (16 host processes) on
262 144 LHS rows of config 4 (Sentinel-2A), config 5 (PROSPECT-PRO, Sentinel-2B) and -- round 6 -- config 4's rows with a per-row
canopy.lidf and canopy.nlayers = 24 (the canopy state SAILH reads from the object, sailh.py:48, 51).  Test infrastructure (imports oracle/); prints one JSON object."""
import json, multiprocessing as mp, os, sys, time
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np

ROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
CASES = (("full", "Sentinel2A-MSI"), ("pro", "Sentinel2B-MSI"), ("canopy_state", "Sentinel2A-MSI"))
NLAYERS = 24          # the "canopy_state" case: config 4's rows with a per-row canopy.lidf (Dirichlet draws) and canopy.nlayers = 24


def case_inputs(kind):
    """(lhs kind, per-row lidf or None, nlayers or None): what both sides evaluate"""
    if kind != "canopy_state":
        return kind, None, None
    return "full", np.random.default_rng(778).dirichlet(np.full(13, 2.0), size=ROWS), NLAYERS


def worker(job):
    kind, sensor, lo, hi = job
    import spart_oracle as O
    from spart_amd import workloads
    T = O.load_tables()
    lk, lidf, nl = case_inputs(kind)
    P = workloads.lhs_params(ROWS, lk, seed=777)[lo:hi]
    out = {k: [] for k in ("R_TOC", "R_TOA", "L_TOA")}
    for i in range(0, len(P), 256):
        kw = {} if lidf is None else dict(lidf=lidf[lo + i:lo + i + 256], nlayers=nl)
        r = O.spart_run(P[i:i + 256], sensor, T, pso="gl", **kw)
        for k in out:
            out[k].append(r[k])
    return {k: np.concatenate(v) for k, v in out.items()}


def main():
    cores = min(16, len(os.sched_getaffinity(0)))
    ref = {}
    with mp.get_context("fork").Pool(cores) as pool:           # before the GPU is touched
        for kind, sensor in CASES:
            t0 = time.time()
            parts = pool.map(worker, [(kind, sensor, i * ROWS // cores, (i + 1) * ROWS // cores) for i in range(cores)])
            ref[kind] = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
            print(f"oracle {kind}: {ROWS} rows in {time.time() - t0:.1f} s on {cores} processes", file=sys.stderr, flush=True)
    import torch
    from spart_amd import get_engine, workloads
    res = {"rows": ROWS, "seed": 777}
    for kind, sensor in CASES:
        lk, lidf, nl = case_inputs(kind)
        P = torch.as_tensor(workloads.lhs_params(ROWS, lk, seed=777).T.copy(), device="cuda:0")
        eng = get_engine(sensor, 0)
        kw = {} if lidf is None else dict(canopy_lidf=torch.as_tensor(lidf, device="cuda:0"), nlayers=nl)
        for dtype in ("float64", "float32"):
            o = eng.run(P, dtype, **kw)
            for k in ("R_TOC", "R_TOA", "L_TOA"):
                x, r = o[k].double().cpu().numpy(), ref[kind][k]
                d = np.abs(x - r)
                floor = 1e-6                                   # SURVEY.md section 8(d)'s metric, both dtypes
                rel = d / np.maximum(np.abs(r), floor)
                res[f"{kind}/{sensor}/{dtype}/{k}"] = {"max_rel": float(rel.max()), "floor": floor, "p99.9_rel": float(np.quantile(rel, 0.999)),
                                                       "max_abs": float(d.max()), "entries_over_tol": int((rel > (1e-6 if dtype == "float64" else 1e-4)).sum()),
                                                       "entries": int(rel.size)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
