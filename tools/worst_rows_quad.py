"""Where does the largest float64 deviation of tools/big_parity.py come from?  HIP float64 columns against the oracle's fast
hot-spot quadrature (pso="gl") on N LHS rows of config 4; the 24 worst rows are then re-evaluated with the oracle on the
reference's own route (scipy QUADPACK for the hot-spot integrals, pso="quad").  Test infrastructure (imports oracle/)."""
import os, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import multiprocessing as mp
import numpy as np

N = int(sys.argv[1]) if len(sys.argv) > 1 else 131072


def worker(job):
    lo, hi = job
    import spart_oracle as O
    from spart_amd import workloads
    T = O.load_tables()
    P = workloads.lhs_params(N, "full", seed=777)[lo:hi]
    return np.concatenate([O.spart_run(P[i:i + 256], "Sentinel2A-MSI", T, pso="gl")["R_TOC"] for i in range(0, len(P), 256)])


if __name__ == "__main__":
    cores = min(16, len(os.sched_getaffinity(0)))
    step = -(-N // (4 * cores))
    with mp.get_context("fork").Pool(cores) as pool:          # (forked before torch / HIP are imported)
        ref = np.concatenate(pool.map(worker, [(i, min(N, i + step)) for i in range(0, N, step)]))
    import torch
    import spart_oracle as O
    from spart_amd import get_engine, workloads
    P = workloads.lhs_params(N, "full", seed=777)
    got = get_engine("Sentinel2A-MSI", 0).run(torch.as_tensor(P.T.copy(), device="cuda:0"), "float64")["R_TOC"].cpu().numpy()
    rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-6)
    rows = np.argsort(rel.max(axis=1))[::-1][:24]
    print(f"{N} rows: max deviation from the oracle with Gauss-Legendre hot-spot integrals {rel.max():.3e}")
    T = O.load_tables()
    q = O.spart_run(P[rows], "Sentinel2A-MSI", T, pso="quad")["R_TOC"]
    relq = np.abs(got[rows] - q) / np.maximum(np.abs(q), 1e-6)
    for r, a, b in zip(rows, rel[rows].max(axis=1), relq.max(axis=1)):
        print(f"  row {r}: vs oracle(gl) {a:.2e}   vs oracle(QUADPACK) {b:.2e}   q = {P[r, 18]:.4f} LAI = {P[r, 15]:.2f}")
    print(f"the 24 worst rows against the QUADPACK route: max {relq.max():.3e}")
    # dissect the worst row: which band, how large is the entry, which stage deviates (full oracle spectra vs materialised HIP spectra)
    r = int(rows[0])
    full = O.spart_run(P[r:r + 1], "Sentinel2A-MSI", T, pso="quad", full=True)
    eng = get_engine("Sentinel2A-MSI", 0)
    fields = ("leaf_refl", "leaf_tran", "soil_refl", "rso", "rdo", "rsd", "rdd")
    hip = eng.run(torch.as_tensor(P[r:r + 1].T.copy(), device="cuda:0"), "float64", materialize=fields)
    j = int(np.argmax(rel[r]))
    print(f"worst row {r}: band {j}, R_TOC oracle {ref[r, j]:.6e} HIP {got[r, j]:.6e} abs diff {abs(got[r, j] - ref[r, j]):.2e}")
    print("   parameters:", dict(zip(workloads.PARAM_NAMES, [float(f'{x:.5g}') for x in P[r]])))
    rho, tau = O.pad_leaf(full["leaf_refl"], full["leaf_tran"])
    exp = dict(leaf_refl=rho, leaf_tran=tau, soil_refl=O.pad_soil(full["soil_refl"]), rso=full["rso"], rdo=full["rdo"], rsd=full["rsd"], rdd=full["rdd"])
    for k in fields:
        a, b = hip[k].cpu().numpy()[0], exp[k][0]
        d = np.abs(a - b)
        i = int(np.argmax(d / np.maximum(np.abs(b), 1e-12)))
        print(f"   {k:10s} max abs {d.max():.2e} at band {int(np.argmax(d))}; max rel {np.max(d / np.maximum(np.abs(b), 1e-12)):.2e} at band {i} (value {b[i]:.3e})")
