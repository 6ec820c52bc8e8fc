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
