"""Look-up-table generation: stream a large parameter table through the GPU in chunks and land the sensor
columns in host memory / on disk.  The host<->device copies of chunk i+1 / i run on their
own HIP streams beside the kernels (double-buffered device buffers, no host staging copy), so the PCIe traffic
(216 B in + 3*nb*4 B out per spectrum) hides behind the evaluation whenever the link keeps up.

On-disk layout (a directory):
    meta.json                  sensor, band ids, band centres, dtype, parameter names, number of rows
    params.npy   (B, 27) f64   the parameter table (workloads.PARAM_NAMES order)
    R_TOC.npy / R_TOA.npy / L_TOA.npy   (B, nb) in the chosen dtype
All .npy files are plain numpy arrays (np.load(..., mmap_mode="r") works for tables larger than RAM).
"""
import ctypes
import json
import os

import numpy as np

from . import workloads
from .engine import get_engine

COLUMNS = ("R_TOC", "R_TOA", "L_TOA")

_MADV_POPULATE_WRITE = 23          # linux/mman.h (Linux >= 5.14): fault the range in, writable, inside ONE system call
_libc = None


def _prefault(a):
    """Make the pages under the C-contiguous numpy view ``a`` resident and writable without touching their contents.
    The destination of a LUT is fresh memory (1.25 GB per 8M Sentinel-2 spectra): left to the download copies, its
    first-touch page faults -- zero-filling 4 KB at a time on the copying thread -- cost more than the kernels that
    produce the data.  Several threads call this on disjoint slices ahead of the pipeline (ctypes releases the GIL).
    Returns False where the kernel does not know MADV_POPULATE_WRITE (then the copies fault the pages in as before)."""
    global _libc
    if a.nbytes == 0:
        return True
    if _libc is None:
        _libc = ctypes.CDLL(None, use_errno=True)
        _libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    page = os.sysconf("SC_PAGE_SIZE")
    lo = a.ctypes.data & ~(page - 1)
    try:
        return _libc.madvise(lo, a.ctypes.data + a.nbytes - lo, _MADV_POPULATE_WRITE) == 0
    except Exception:               # noqa: BLE001  (pre-faulting is an optimisation: never let it fail the LUT)
        return False


class LutBlock(dict):
    """Result of generate_lut: the column arrays, with ``rows = (lo, hi)`` = the rows THIS process evaluated (the whole
    table unless ``shard=True`` under a process group) and ``total`` = the table's row count.  Without ``path`` a sharded call
    returns arrays holding only the rows lo..hi; with ``path`` the arrays are memmaps of the whole files."""
    rows = (0, 0)
    total = 0


def _group_info(shard, group):
    """(world, rank) of the process group a sharded call runs under; (1, 0) for a plain call"""
    if not shard:
        return 1, 0
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def open_lut_files(path, files, world=1, rank=0, group=None):
    """The .npy files of a LUT directory as writable memmaps shared by the ranks of a sharded generate_lut.
    files: [(name, dtype, shape)].  Rank 0 creates them at their final size (and removes a manifest left over from an earlier
    table: meta.json is what says "complete"); after ONE barrier the other ranks open the same files read-write; every rank
    then writes only its own rows (single node: one page cache; on a network file system every rank must flush before the
    closing barrier, which generate_lut does)."""
    whole = None
    if rank == 0:
        os.makedirs(path, exist_ok=True)
        stale = os.path.join(path, "meta.json")
        if os.path.exists(stale):
            os.remove(stale)
        whole = {k: np.lib.format.open_memmap(os.path.join(path, k + ".npy"), mode="w+", dtype=dt_, shape=shp)
                 for k, dt_, shp in files}
    if world > 1:
        import torch.distributed as dist
        dist.barrier(group)                        # the files exist with their final size
    if rank != 0:
        whole = {k: np.lib.format.open_memmap(os.path.join(path, k + ".npy"), mode="r+") for k, _, _ in files}
        for k, dt_, shp in files:
            if whole[k].shape != tuple(shp) or whole[k].dtype != np.dtype(dt_):
                raise RuntimeError(f"{path}/{k}.npy is {whole[k].shape} {whole[k].dtype}, expected {tuple(shp)} {np.dtype(dt_)}")
    return whole


def generate_lut(params, sensor, path=None, dtype="float32", chunk=1 << 18, device=None, prune=True, fault_threads=8,
                 f32_bands=False, shard=False, group=None, out=None):
    """params: (B, 27) array-like on the HOST (numpy / memmap).  Returns dict of host arrays (np.memmap when
    ``path`` is given).

    ``shard=True`` under an initialised torch.distributed process group (one process per GPU): the table is cut into
    contiguous blocks of ceil(B / world) rows (sharding.shard_bounds) and every rank evaluates ITS block on its own device
    and writes it at its own rows -- with ``path``, straight into the directory's R_TOC.npy / R_TOA.npy / L_TOA.npy /
    params.npy (rank 0 creates the files, the other ranks open them read-write after ONE barrier; a second barrier, then rank
    0 writes meta.json).  No result ever crosses ranks: there is no collective on the data path, and the directory is
    byte-identical to a single-process run.  Without ``path`` every rank gets the arrays of its own block (LutBlock.rows).

    ``prune=True`` (default: a LUT holds the sensor columns only) evaluates just the <= 2 nb bands those columns
    depend on -- bit-identical columns; ``prune=False`` also evaluates the other bands of every spectrum (band sums);
    ``dtype="float64", f32_bands=True`` gives float64 columns identical to the float64 mode's at the float32 mode's speed.
    ``out``: optional dict of caller-owned host arrays for the three columns (this rank's rows), e.g. a previous call's result:
    their pages are already resident.  Fresh arrays cost 1.25 GB of first-touch page faults per 8M spectra even with the helper
    threads -- 8M pruned: 57 ms fresh, 40 ms reused (2.0e8 spectra/s; the two PCIe directions alone need 33 ms);
    transparent huge pages made it worse on the test box (madvise mode with direct compaction: 71 ms), so the arrays are ordinary.

    Pipeline per chunk i (three HIP streams; the host never holds a private staging copy):
        upload(i+1)   H2D of the next (n, 27) rows straight from the caller's table + on-device transpose to the
                      structure-of-arrays layout the kernels read          -- overlaps the kernels of chunk i
        launch(i+1)   queued behind chunk i on the compute stream
        download(i)   D2H of the three (n, nb) column blocks straight into the destination arrays / memmaps
                                                                            -- overlaps the kernels of chunk i+1
        prefault(i+2) ``fault_threads`` helper threads make the destination pages of chunk i+2 resident (see _prefault)
    ``chunk`` = 262 144 rows by default: the pipeline's fill (first upload) and drain (last download) are not overlapped
    with anything, so smaller chunks waste less (8M spectra, all bands: 112 ms with 1M-row chunks, 105 ms with 256k; below
    that the per-chunk launch / copy overheads win).
    The copies block the HOST thread (pageable memory) but not the GPU, which stays busy as long as the two copies
    of a chunk (372 B per spectrum, ~7 ms per 1M at PCIe Gen5 rates) take less than its kernels (12 ms per 1M)."""
    import warnings
    from concurrent.futures import ThreadPoolExecutor

    import torch

    from .sharding import shard_bounds
    P = np.asarray(params) if not isinstance(params, np.memmap) else params
    if P.ndim != 2 or P.shape[1] != workloads.NPARAM:
        raise ValueError("params must be (B, 27)")
    Btot = P.shape[0]
    world, rank = _group_info(shard, group)
    lo0, hi0 = shard_bounds(Btot, world, rank)
    eng = get_engine(sensor, device)
    nb = eng.nb
    npdt = np.float32 if dtype in ("float32", "fp32", "f32") else np.float64
    tdt = torch.float32 if npdt is np.float32 else torch.float64
    if path is not None and out is not None:
        raise ValueError("generate_lut: give `path` (results land in the directory's .npy files) or `out` (caller-owned arrays), not both")
    if path is not None:
        whole = open_lut_files(path, [(k, npdt, (Btot, nb)) for k in COLUMNS] + [("params", np.float64, tuple(P.shape))],
                               world, rank, group)
        full_out = {k: whole[k] for k in COLUMNS}
        out = {k: whole[k][lo0:hi0] for k in COLUMNS}          # this rank's rows of the shared files
        pm = whole["params"][lo0:hi0]
    else:
        full_out = None
        if out is not None:                                    # caller-owned destination (reused between tables: its pages are resident)
            for k in COLUMNS:
                a = out.get(k)
                if not isinstance(a, np.ndarray) or a.shape != (hi0 - lo0, nb) or a.dtype != npdt or not a.flags.c_contiguous or not a.flags.writeable:
                    raise ValueError(f"out[{k!r}] must be a writable C-contiguous ({hi0 - lo0}, {nb}) {np.dtype(npdt).name} array")
            out = {k: out[k] for k in COLUMNS}
        else:
            out = {k: np.empty((hi0 - lo0, nb), dtype=npdt) for k in COLUMNS}
        pm = None
    P = P[lo0:hi0]
    B = hi0 - lo0
    chunk = int(max(1, min(chunk, max(B, 1))))
    dev = eng.device
    compute = torch.cuda.current_stream(dev)
    h2d, d2h = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    drow = [torch.empty((chunk, workloads.NPARAM), dtype=torch.float64, device=dev) for _ in range(2)]   # as on the host
    dsoa = [torch.empty((workloads.NPARAM, chunk), dtype=torch.float64, device=dev) for _ in range(2)]   # kernel layout
    dout = [torch.empty((3, chunk, nb), dtype=tdt, device=dev) for _ in range(2)]
    ev_in = [torch.cuda.Event() for _ in range(2)]
    ev_done = [torch.cuda.Event() for _ in range(2)]
    nchunks = (B + chunk - 1) // chunk
    keep = [None, None]         # host source of the upload in flight on buffer j (a temporary when P needed converting)

    def bounds(i):
        lo = i * chunk
        return lo, min(chunk, B - lo)

    def as_tensor(a):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")          # read-only inputs (memmaps opened "r") are only read
            return torch.from_numpy(a)

    def upload(i):
        j = i % 2
        lo, n = bounds(i)
        src = P[lo:lo + n]
        if src.dtype != np.float64 or not src.flags.c_contiguous:
            src = np.ascontiguousarray(src, dtype=np.float64)
        with torch.cuda.stream(h2d):
            if i >= 2:
                h2d.wait_event(ev_done[j])                         # dsoa[j] is free once chunk i-2's kernels ran
            keep[j] = src
            drow[j][:n].copy_(as_tensor(src), non_blocking=True)
            dsoa[j][:, :n].copy_(drow[j][:n].t())                  # -> structure of arrays (27, n)
            ev_in[j].record(h2d)
        if pm is not None:
            pm[lo:lo + n] = src

    def launch(i):
        j = i % 2
        lo, n = bounds(i)
        compute.wait_event(ev_in[j])
        res = dout[j][:, :n]
        Pd = dsoa[j] if n == chunk else dsoa[j][:, :n].contiguous()
        eng.run(Pd, dtype, out={"R_TOC": res[0], "R_TOA": res[1], "L_TOA": res[2]}, prune=prune, f32_bands=f32_bands)
        ev_done[j].record(compute)

    fault_threads = max(0, int(fault_threads))
    fpool = ThreadPoolExecutor(fault_threads) if fault_threads else None
    faults = {}                                                    # chunk -> futures of its prefault tasks

    def prefault(i):
        if fpool is None or i >= nchunks or i in faults:
            return
        lo, n = bounds(i)
        dests = [out[k][lo:lo + n] for k in COLUMNS] + ([pm[lo:lo + n]] if pm is not None else [])
        per = max(1, -(-n // max(1, fault_threads // len(dests))))
        faults[i] = [fpool.submit(_prefault, d[r:r + per]) for d in dests for r in range(0, n, per)]

    def download(i):                                               # runs on the helper thread
        j = i % 2
        lo, n = bounds(i)
        for f in faults.pop(i, ()):
            try:
                f.result()                                         # the destination pages of this chunk are resident
            except Exception:       # noqa: BLE001  ("not pre-faulted": the copy below faults the pages in itself)
                pass
        with torch.cuda.device(dev), torch.cuda.stream(d2h):
            d2h.wait_event(ev_done[j])
            for q, k in enumerate(COLUMNS):
                as_tensor(out[k][lo:lo + n]).copy_(dout[j][q, :n], non_blocking=True)
            d2h.synchronize()                                      # dout[j] may be overwritten by chunk i+2

    # downloads (which also take the first-touch page faults of the destination) run on one helper thread so that
    # they overlap the uploads issued by this thread; torch releases the GIL inside the copies
    fut = [None, None]
    try:
        with ThreadPoolExecutor(1) as pool:
            prefault(0)
            prefault(1)
            if nchunks:
                upload(0)
                launch(0)
            for i in range(nchunks):
                prefault(i + 2)
                if i + 1 < nchunks:
                    upload(i + 1)                                  # overlaps the kernels of chunk i
                    if fut[(i + 1) % 2] is not None:
                        fut[(i + 1) % 2].result()                  # dout[(i+1) % 2] has been drained (chunk i-1)
                    launch(i + 1)
                fut[i % 2] = pool.submit(download, i)              # overlaps the kernels of chunk i+1
            for f in fut:
                if f is not None:
                    f.result()
    finally:                                                       # also on an exception in upload / launch / download
        if fpool is not None:
            fpool.shutdown(wait=True, cancel_futures=True)
    if path is not None:
        for a in whole.values():
            a.flush()
        if world > 1:
            import torch.distributed as dist
            dist.barrier(group)                    # every rank's rows are in the files
        if rank == 0:
            meta = {"sensor": sensor, "bands": list(eng.band_id), "wavelengths": [float(w) for w in eng.wl_smac],
                    "dtype": np.dtype(npdt).name, "rows": int(Btot), "param_names": workloads.PARAM_NAMES,
                    "columns": list(COLUMNS), "pruned": bool(prune)}
            with open(os.path.join(path, "meta.json"), "w") as f:
                json.dump(meta, f, indent=1)
    res = LutBlock(full_out if full_out is not None else out)
    res.rows, res.total = (int(lo0), int(hi0)), int(Btot)
    return res


def load_lut(path, mmap=True):
    """-> (meta dict, params, dict of columns), memory-mapped by default."""
    with open(os.path.join(path, "meta.json")) as f:
        meta = json.load(f)
    mode = "r" if mmap else None
    cols = {k: np.load(os.path.join(path, k + ".npy"), mmap_mode=mode) for k in meta["columns"]}
    return meta, np.load(os.path.join(path, "params.npy"), mmap_mode=mode), cols


def invert_lut(lut, obs, column="R_TOC", weights=None, dtype=None, shard=False, group=None, device=None, stats=False):
    """Nearest LUT row per observed spectrum (spart_lut_nearest: exact argmin of the weighted squared distance, lowest index
    on ties) over a LUT directory written by generate_lut (``lut`` = its path) or an in-memory (B, nb) array.

    ``shard=True`` under a process group: every rank searches ITS contiguous block of rows (read from the directory's memmap:
    only those rows are ever touched) and ONE all_gather of (cost, global row) per observation settles the winner
    (sharding.lut_nearest_sharded); every rank returns the same (idx (M,) int64 numpy, cost (M,) numpy), equal bit for bit to
    the single-process search.  ``obs`` must be the same on every rank."""
    import torch
    from .sharding import lut_nearest_sharded, shard_bounds
    if isinstance(lut, (str, os.PathLike)):
        meta, _, cols = load_lut(lut)
        table = cols[column]
        dtype = dtype or meta["dtype"]
    else:
        table = lut
        dtype = dtype or ("float64" if np.asarray(lut[:1]).dtype == np.float64 else "float32")
    world, rank = _group_info(shard, group)
    lo, hi = shard_bounds(table.shape[0], world, rank)
    eng = get_engine(None, device)
    td = torch.float32 if dtype in ("float32", "fp32", "f32") else torch.float64
    local = torch.as_tensor(np.array(table[lo:hi])).to(device=eng.device, dtype=td)      # (a copy: memmaps opened read-only)
    o = torch.as_tensor(np.asarray(obs)).to(device=eng.device, dtype=td)
    info = {}

    def nearest(l, ob):
        r = eng.lut_nearest(l, ob, weights=weights, dtype=dtype, stats=stats)
        if stats:
            info.update(r[2])
        return r[0], r[1]
    comm = None
    if world > 1:
        import torch.distributed as dist
        if dist.get_backend(group) != "nccl":          # gloo moves host tensors: the (M, 2) winners travel through the host
            comm = "cpu"
    idx, cost = lut_nearest_sharded(local, lo, o, nearest, group if world > 1 else None, comm_device=comm)
    out = (idx.cpu().numpy(), cost.cpu().numpy())
    return out + (dict(info, rows=(lo, hi)),) if stats else out


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
