"""One-off, container-only: oracle vs the REAL reference on extra LHS rows (beyond tests/golden/e2e.npz).

    python tools/validate_oracle_large.py [rows_scale]

Runs the reference's SPART(...).run() (tests/golden/make_golden.py: run_row) on 8 worker processes for
fresh Latin-hypercube rows (seed 4242, not the fixture seed) and prints the max relative difference of
oracle/spart_oracle.py per output column.  Nothing is stored; the numbers are quoted in DESIGN.md.
"""
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import make_golden as mg  # noqa: E402  (imports the reference; generators only run under __main__)
import spart_oracle as O  # noqa: E402

CASES = (("full", "Sentinel2A-MSI", 1024), ("pro", "Sentinel2B-MSI", 512),
         ("full", "TerraAqua-MODIS", 512), ("full", "Sentinel3A-OLCI", 256), ("full", "LANDSAT8-OLI", 256))


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    T = O.load_tables()
    for kind, sensor, n in CASES:
        n = max(8, int(n * scale))
        P = mg.workloads.lhs_params(n, kind, seed=4242)
        t0 = time.time()
        with Pool(8) as pool:
            res = pool.map(mg.run_row, [(r, sensor) for r in P], chunksize=8)
        ref = {k: np.array([r[j] for r in res]) for j, k in enumerate(("R_TOC", "R_TOA", "L_TOA"))}
        o = O.spart_run(P, sensor, T, pso="gl")
        d = {}
        for k in ref:
            fin = np.isfinite(ref[k])
            d[k] = "%.2e" % np.max(np.abs(o[k][fin] - ref[k][fin]) / np.maximum(np.abs(ref[k][fin]), 1e-6))
        nan_mismatch = sum(int(np.sum(np.isfinite(ref[k]) != np.isfinite(o[k]))) for k in ref)
        print(f"{kind:4s} {sensor:16s} rows={n:5d} {time.time() - t0:5.0f}s  max rel diff {d}  non-finite mismatches {nan_mismatch}",
              flush=True)


if __name__ == "__main__":
    main()
