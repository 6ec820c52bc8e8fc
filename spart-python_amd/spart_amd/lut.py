"""Look-up-table generation: stream a large parameter table through the GPU in chunks and land the sensor
columns in host memory / on disk.  The host<->device copies of chunk i+1 / i-1 run on their
own HIP streams beside the kernels of chunk i (double-buffered pinned staging), so the PCIe traffic
(216 B in + 3*nb*4 B out per spectrum) hides behind the evaluation whenever the link keeps up.

On-disk layout (a directory):
    meta.json                  sensor, band ids, band centres, dtype, parameter names, number of rows
    params.npy   (B, 27) f64   the parameter table (workloads.PARAM_NAMES order)
    R_TOC.npy / R_TOA.npy / L_TOA.npy   (B, nb) in the chosen dtype
All .npy files are plain numpy arrays (np.load(..., mmap_mode="r") works for tables larger than RAM).
"""
import json
import os

import numpy as np

from . import workloads
from .engine import get_engine

COLUMNS = ("R_TOC", "R_TOA", "L_TOA")


def generate_lut(params, sensor, path=None, dtype="float32", chunk=1 << 20, device=None, prune=False):
    """params: (B, 27) array-like on the HOST (numpy / memmap).  Returns dict of host arrays (np.memmap when
    ``path`` is given).  ``prune=True`` evaluates only the bands the sensor needs (identical columns)."""
    import torch

    P = np.asarray(params) if not isinstance(params, np.memmap) else params
    if P.ndim != 2 or P.shape[1] != workloads.NPARAM:
        raise ValueError("params must be (B, 27)")
    B = P.shape[0]
    eng = get_engine(sensor, device)
    nb = eng.nb
    npdt = np.float32 if dtype in ("float32", "fp32", "f32") else np.float64
    tdt = torch.float32 if npdt is np.float32 else torch.float64
    if path is not None:
        os.makedirs(path, exist_ok=True)
        out = {k: np.lib.format.open_memmap(os.path.join(path, k + ".npy"), mode="w+", dtype=npdt, shape=(B, nb))
               for k in COLUMNS}
        pm = np.lib.format.open_memmap(os.path.join(path, "params.npy"), mode="w+", dtype=np.float64, shape=P.shape)
    else:
        out = {k: np.empty((B, nb), dtype=npdt) for k in COLUMNS}
        pm = None
    chunk = int(max(1, min(chunk, max(B, 1))))
    dev = eng.device
    compute = torch.cuda.current_stream(dev)
    h2d, d2h = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    # double-buffered staging: pinned host + device, for parameters in and columns out
    hin = [torch.empty((chunk, workloads.NPARAM), dtype=torch.float64).pin_memory() for _ in range(2)]
    din = [torch.empty((workloads.NPARAM, chunk), dtype=torch.float64, device=dev) for _ in range(2)]
    dout = [torch.empty((3, chunk, nb), dtype=tdt, device=dev) for _ in range(2)]
    hout = [torch.empty((3, chunk, nb), dtype=tdt).pin_memory() for _ in range(2)]
    ev_in = [torch.cuda.Event() for _ in range(2)]
    ev_done = [torch.cuda.Event() for _ in range(2)]
    ev_out = [torch.cuda.Event() for _ in range(2)]
    ev_free = [torch.cuda.Event() for _ in range(2)]
    pending = [None, None]          # (lo, n) whose columns sit in hout[j] once ev_out[j] has passed
    nchunks = (B + chunk - 1) // chunk

    def stage_in(i):
        j = i % 2
        lo = i * chunk
        n = min(chunk, B - lo)
        hin[j][:n].copy_(torch.from_numpy(np.ascontiguousarray(P[lo:lo + n], dtype=np.float64)))
        if pm is not None:
            pm[lo:lo + n] = hin[j][:n].numpy()
        with torch.cuda.stream(h2d):
            h2d.wait_event(ev_free[j]) if i >= 2 else None        # din[j] is free once chunk i-2's kernels ran
            tmp = hin[j][:n].to(dev, non_blocking=True)           # (n, 27)
            din[j][:, :n].copy_(tmp.t())                          # -> structure of arrays (27, n)
            ev_in[j].record(h2d)
        return lo, n

    def drain(j):
        if pending[j] is not None:
            lo, n = pending[j]
            ev_out[j].synchronize()
            for q, k in enumerate(COLUMNS):
                out[k][lo:lo + n] = hout[j][q, :n].numpy()
            pending[j] = None

    if nchunks:
        nxt = stage_in(0)
    for i in range(nchunks):
        j = i % 2
        lo, n = nxt
        if i + 1 < nchunks:
            nxt = stage_in(i + 1)                                  # overlaps with the kernels below
        drain(j)                                                   # hout[j] / dout[j] from chunk i-2 must be consumed
        compute.wait_event(ev_in[j])
        res = dout[j][:, :n]
        if din[j].shape[1] == n:
            Pd = din[j]
        else:
            Pd = din[j][:, :n].contiguous()
        eng.run(Pd, dtype, out={"R_TOC": res[0], "R_TOA": res[1], "L_TOA": res[2]}, prune=prune)
        ev_done[j].record(compute)
        ev_free[j].record(compute)
        with torch.cuda.stream(d2h):
            d2h.wait_event(ev_done[j])
            hout[j][:, :n].copy_(res, non_blocking=True)
            ev_out[j].record(d2h)
        pending[j] = (lo, n)
    drain(0)
    drain(1)
    if path is not None:
        for a in out.values():
            a.flush()
        pm.flush()
        meta = {"sensor": sensor, "bands": list(eng.band_id), "wavelengths": [float(w) for w in eng.wl_smac],
                "dtype": np.dtype(npdt).name, "rows": int(B), "param_names": workloads.PARAM_NAMES,
                "columns": list(COLUMNS), "pruned": bool(prune)}
        with open(os.path.join(path, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1)
    return out


def load_lut(path, mmap=True):
    """-> (meta dict, params, dict of columns), memory-mapped by default."""
    with open(os.path.join(path, "meta.json")) as f:
        meta = json.load(f)
    mode = "r" if mmap else None
    cols = {k: np.load(os.path.join(path, k + ".npy"), mmap_mode=mode) for k in meta["columns"]}
    return meta, np.load(os.path.join(path, "params.npy"), mmap_mode=mode), cols


def lut_to_parquet(path, parquet_path, compression="gzip"):
    """One wide table (parameters + <column>_<band centre>) like the reference's golden files
    (tests/unit/test_PROSPECT/build_PROSPECT_tests.py:35); for LUTs that fit in memory."""
    import pandas as pd
    meta, params, cols = load_lut(path)
    df = pd.DataFrame(np.asarray(params), columns=meta["param_names"])
    for k in meta["columns"]:
        for j, w in enumerate(meta["wavelengths"]):
            df[f"{k}_{w:g}"] = np.asarray(cols[k][:, j])
    df.to_parquet(parquet_path, compression=compression)
    return parquet_path
