"""Child process of tests/test_gpu_parity.py::test_parity_at_scale_against_the_oracle: evaluates the ORACLE (oracle/spart_oracle.py,
numpy float64, test infrastructure) on fresh Latin-hypercube rows with a pool of host processes and writes the three sensor
columns to an .npz.  A separate program because the pytest process has initialised HIP by then and must not fork; this one
never touches the GPU.

    python tests/helpers/oracle_rows.py <kind: full|pro> <sensor> <rows> <seed> <out.npz> [processes]
"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd", "spart_amd"))

_S = {}


def _init(kind, sensor, rows, seed):
    import spart_oracle as O
    import workloads                     # plain module: no package import, no HIP library
    _S.update(O=O, T=O.load_tables(), P=workloads.lhs_params(rows, kind, seed=seed), sensor=sensor)


def _block(span):
    lo, hi = span
    with np.errstate(all="ignore"):
        r = _S["O"].spart_run(_S["P"][lo:hi], _S["sensor"], _S["T"], pso="gl")
    return {k: r[k] for k in ("R_TOC", "R_TOA", "L_TOA")}


def main():
    kind, sensor, rows, seed, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    nproc = int(sys.argv[6]) if len(sys.argv) > 6 else min(16, len(os.sched_getaffinity(0)))
    t0 = time.time()
    spans = [(i, min(i + 256, rows)) for i in range(0, rows, 256)]
    with mp.get_context("fork").Pool(nproc, initializer=_init, initargs=(kind, sensor, rows, seed)) as pool:
        parts = pool.map(_block, spans, chunksize=1)
    np.savez(out, seconds=time.time() - t0, processes=nproc, **{k: np.concatenate([p[k] for p in parts]) for k in parts[0]})


if __name__ == "__main__":
    main()
