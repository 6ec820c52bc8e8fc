"""world_size-2 (and 3, ragged) run of the sharded path on CPU with the gloo backend.  The HIP
evaluator is replaced by the oracle (tests may use it); what is under test is the shard
arithmetic and the single gather."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _load_sharding():
    import importlib.util
    p = os.path.join(ROOT, "spart-python_amd", "spart_amd", "sharding.py")
    spec = importlib.util.spec_from_file_location("_sharding", p)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _worker(rank, world, port, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, HERE)
    import spart_oracle as O
    import spart_amd_workloads as W
    sh = _load_sharding()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T = O.load_tables()
    P = torch.as_tensor(W.lhs_params(B, "full", seed=9).T.copy())

    def evaluate(Ps):
        if Ps.shape[1] == 0:
            return torch.zeros((3, 0, 13), dtype=torch.float64)
        o = O.spart_run(Ps.numpy().T, "Sentinel2A-MSI", T, pso="gl")
        return torch.as_tensor(np.stack([o["R_TOC"], o["R_TOA"], o["L_TOA"]]))

    res = sh.run_sharded(P, evaluate)
    if rank == 0:
        full = evaluate(P)
        q.put((tuple(res.shape), float((res - full).abs().max())))
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,B", [(2, 24), (3, 10), (2, 1)])
def test_sharded_gather_matches_single(world, B):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    shape, err = q.get(timeout=240)
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    assert shape == (3, B, 13)
    assert err == 0.0


def test_shard_bounds():
    sh = _load_sharding()
    for B, w in ((1_000_000, 8), (10, 3), (1, 2), (0, 4), (7, 7), (5, 8)):
        cover = []
        for r in range(w):
            lo, hi = sh.shard_bounds(B, w, r)
            assert 0 <= lo <= hi <= B
            cover += list(range(lo, hi))
        assert cover == list(range(B))
    assert sh.shard_bounds(1_000_000, 8, 3) == (375_000, 500_000)
    with pytest.raises(ValueError):
        sh.shard_bounds(10, 2, 2)


# ------------------------------------------------------------------------------------------------ LUT rows (SURVEY 8e x 8f)
def _lut_case(B, nb, M, seed):
    """a LUT with exact duplicates (ties across and inside shards), a row of NaN, and observations that hit rows exactly"""
    rng = np.random.default_rng(seed)
    lut = rng.uniform(0.0, 0.6, (B, nb)).astype(np.float32)
    if B > 6:
        lut[B - 2] = lut[1]                    # the same spectrum in the first and the last shard: the LOWER index must win
        lut[B // 2] = lut[B // 2 - 1]          # ... and next to each other
        lut[3] = np.nan                        # never wins
    obs = (lut[rng.integers(0, B, M)] * (1 + 0.02 * rng.standard_normal((M, nb)))).astype(np.float32)
    obs[np.isnan(obs)] = 0.25
    if B > 6:
        obs[0] = lut[1]
        obs[1] = lut[B // 2]
    return lut, obs


def _lut_worker(rank, world, port, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from lut_brute_force import brute_force_numpy
    sh = _load_sharding()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lut, obs = _lut_case(B, 13, 40, seed=5)
    w = np.linspace(0.5, 1.5, 13).astype(np.float32)
    calls = []

    def nearest(l, o):                          # the exact single-device search, here the brute force of the defined cost
        calls.append(l.shape[0])
        i, c = brute_force_numpy(l.numpy(), o.numpy(), w)
        return torch.as_tensor(i), torch.as_tensor(c)

    lo, hi = sh.shard_bounds(B, world, rank)
    idx, cost = sh.lut_nearest_sharded(torch.as_tensor(lut[lo:hi]), lo, torch.as_tensor(obs), nearest)
    ti, tc = brute_force_numpy(lut, obs, w)
    same = bool(np.array_equal(idx.numpy(), ti) and np.array_equal(cost.numpy(), tc))       # on EVERY rank
    # float64 costs travel as their own bit pattern too; an all-NaN LUT has no winner anywhere
    i64, c64 = sh.lut_nearest_sharded(torch.as_tensor(lut[lo:hi].astype(np.float64)), lo, torch.as_tensor(obs.astype(np.float64)),
                                      lambda l, o: tuple(torch.as_tensor(x) for x in brute_force_numpy(l.numpy(), o.numpy())))
    t64 = brute_force_numpy(lut.astype(np.float64), obs.astype(np.float64))
    same64 = bool(np.array_equal(i64.numpy(), t64[0]) and np.array_equal(c64.numpy(), t64[1]))
    nan_lut = np.full_like(lut, np.nan)
    ni, nc = sh.lut_nearest_sharded(torch.as_tensor(nan_lut[lo:hi]), lo, torch.as_tensor(obs),
                                    lambda l, o: tuple(torch.as_tensor(x) for x in brute_force_numpy(l.numpy(), o.numpy())))
    none = bool((ni == -1).all() and torch.isinf(nc).all())
    q.put((rank, same, same64, none, len(calls) == (1 if hi > lo else 0), int(ti[0]), int(ti[1])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,B", [(2, 101), (3, 50), (3, 2)])
def test_lut_rows_sharded_nearest_matches_single_search(world, B):
    """LUT inversion with the LUT row-sharded: per-rank exact search + ONE all_gather of (cost bits, global row) -> the same
    winners and costs as one search over the whole LUT, bit for bit, on every rank; ties go to the lowest GLOBAL row (across
    shards too), NaN rows never win, ranks with an empty block (B < world) take part in the collective without searching."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_lut_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    assert [g[0] for g in got] == list(range(world))
    for g in got:
        assert g[1] and g[2] and g[3] and g[4], g
    if B > 6:
        assert got[0][5] == 1 and got[0][6] == B // 2 - 1           # duplicates: first / lower row wins, whichever shard holds it


def test_select_nearest_rule():
    sh = _load_sharding()
    inf = float("inf")
    costs = torch.tensor([[1.0, 2.0, inf, 5.0], [1.0, 1.5, inf, 5.0], [0.5, 1.5, inf, inf]])
    idxs = torch.tensor([[7, 3, -1, 9], [2, 8, -1, 4], [11, 6, -1, -1]])
    i, c = sh.select_nearest(costs, idxs)
    assert i.tolist() == [11, 6, -1, 4] and c.tolist() == [0.5, 1.5, inf, 5.0]


def _files_worker(rank, world, port, path, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib.util
    sh = _load_sharding()
    # lut.py's file handling only (generate_lut itself needs the HIP engine: tests/test_gpu_sharded.py)
    src = open(os.path.join(ROOT, "spart-python_amd", "spart_amd", "lut.py")).read()
    ns = {"__name__": "_lutfiles"}
    start, end = src.index("def open_lut_files("), src.index("def generate_lut(")
    exec("import os\nimport numpy as np\n" + src[start:end], ns)
    files = [("R_TOC", np.float32, (B, 13)), ("params", np.float64, (B, 27))]
    whole = ns["open_lut_files"](path, files, world, rank, None)
    lo, hi = sh.shard_bounds(B, world, rank)
    whole["R_TOC"][lo:hi] = np.arange(lo * 13, hi * 13, dtype=np.float32).reshape(-1, 13)
    whole["params"][lo:hi] = np.arange(lo * 27, hi * 27, dtype=np.float64).reshape(-1, 27)
    for a in whole.values():
        a.flush()
    dist.barrier()
    if rank == 0:
        a = np.load(os.path.join(path, "R_TOC.npy"))
        p = np.load(os.path.join(path, "params.npy"))
        q.put((bool(np.array_equal(a, np.arange(B * 13, dtype=np.float32).reshape(B, 13))),
               bool(np.array_equal(p, np.arange(B * 27, dtype=np.float64).reshape(B, 27))), os.path.exists(os.path.join(path, "meta.json"))))
    dist.barrier()
    dist.destroy_process_group()


def test_lut_directory_written_by_three_ranks(tmp_path):
    """every rank writes its own rows of the SAME .npy files (rank 0 creates them, one barrier, the others open read-write);
    a manifest left over from an earlier table is removed before anything is written"""
    path = str(tmp_path / "lut")
    os.makedirs(path)
    open(os.path.join(path, "meta.json"), "w").write("{}")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port, B, world = _free_port(), 1001, 3
    procs = [ctx.Process(target=_files_worker, args=(r, world, port, path, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok_cols, ok_params, stale = q.get(timeout=240)
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    assert ok_cols and ok_params and not stale
