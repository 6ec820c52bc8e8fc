"""The HIP + collective leg of the sharded path (spart_amd.sharding) on the one GPU of the test box:

* a world-size-1 `nccl` (= RCCL) process group: run_sharded(P, default_evaluate(...)) -- the real evaluator and a
  real RCCL gather of device tensors -- against the reference's golden rows of the config-4 LHS;
* two ranks under `gloo`, BOTH on device 0, each running the HIP evaluator on its shard (ragged: 257 rows), the
  gathered result compared bit for bit with a single-rank evaluation of the whole batch.

8-GPU RCCL runs are the driver's (bench.py --gpus 8); nothing here claims them."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _nccl_world1(port, q):
    """own process: a process group cannot be re-initialised inside the pytest process once destroyed elsewhere"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
    import torch
    import torch.distributed as dist
    from spart_amd import sharding
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    g = np.load(os.path.join(HERE, "golden", "e2e.npz"))
    P = torch.as_tensor(g["lhs_full/Sentinel2A-MSI/P"].T.copy(), device="cuda:0")
    out = {}
    for dtype in ("float64", "float32"):
        res = sharding.run_sharded(P, sharding.default_evaluate("Sentinel2A-MSI", dtype, 0))
        torch.cuda.synchronize()
        out[dtype] = res.double().cpu().numpy()
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_hip_path_under_nccl_world1(golden):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_world1, args=(_free_port(), q))
    p.start()
    out = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    g = golden["e2e"]
    for dtype, tol in (("float64", 1e-6), ("float32", 1e-4)):
        assert out[dtype].shape == (3, 256, 13)
        for i, k in enumerate(("R_TOC", "R_TOA", "L_TOA")):
            assert rel_err(out[dtype][i], g[f"lhs_full/Sentinel2A-MSI/{k}"], 1e-6) < tol, (dtype, k)


def _gloo_hip_rank(rank, world, port, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
    import torch
    import torch.distributed as dist
    from spart_amd import sharding, workloads
    torch.cuda.set_device(0)                                        # both ranks share the one GPU
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P = torch.as_tensor(workloads.lhs_params(B, "full", seed=12).T.copy())
    ev = sharding.default_evaluate("Sentinel2A-MSI", "float32", 0)

    def evaluate(Ps):                                               # HIP evaluator, result handed to gloo on the host
        return ev(Ps.to("cuda:0")).cpu()

    res = sharding.run_sharded(P, evaluate)
    if rank == 0:
        full = evaluate(P)
        q.put((tuple(res.shape), bool(torch.equal(res, full)), bool(torch.isfinite(res).all())))
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_gloo_ranks_with_the_hip_evaluator_match_single_rank():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port, B = _free_port(), 257
    procs = [ctx.Process(target=_gloo_hip_rank, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    shape, same, finite = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert shape == (3, B, 13) and finite
    assert same                                                    # shards are independent: bit-identical to one rank


def test_eight_simulated_shards_on_one_gpu():
    """the 8-way split of BASELINE config 4 rehearsed on ONE device, no process group: the eight contiguous shards of
    sharding.shard_bounds evaluated one after the other and concatenated are bit-identical to the single evaluation (ragged last
    shard), and the LUT inversion over eight row blocks -- the per-block winners stacked as the all_gather would stack them, then
    sharding.select_nearest -- returns the single search's winners and costs."""
    import torch
    from spart_amd import get_engine, sharding, workloads
    eng = get_engine("Sentinel2A-MSI", 0)
    B, G = 200_003, 8
    P = torch.as_tensor(workloads.lhs_params(B, "full", seed=5).T.copy(), device="cuda:0")
    whole = {k: v.clone() for k, v in eng.run(P, "float32").items()}
    parts = {k: [] for k in whole}
    rows = 0
    for r in range(G):
        lo, hi = sharding.shard_bounds(B, G, r)
        o = eng.run(P[:, lo:hi].contiguous(), "float32")
        for k in parts:
            parts[k].append(o[k].clone())
        rows += hi - lo
    assert rows == B and sharding.shard_bounds(B, G, 7) == (175_007, 200_003)
    for k in whole:
        assert torch.equal(torch.cat(parts[k]), whole[k]), k
    lut = whole["R_TOC"]
    g = torch.Generator(device="cuda:0").manual_seed(2)
    obs = lut[torch.randint(0, B, (4096,), generator=g, device="cuda:0")] * (1 + 0.02 * torch.randn((4096, 13), generator=g, device="cuda:0"))
    si, sc = eng.lut_nearest(lut, obs)
    costs, idxs = [], []
    for r in range(G):
        lo, hi = sharding.shard_bounds(B, G, r)
        i, c = sharding.lut_nearest_sharded(lut[lo:hi], lo, obs, eng.lut_nearest)      # (no group: the local search with global rows)
        costs.append(c)
        idxs.append(i)
    bi, bc = sharding.select_nearest(torch.stack(costs), torch.stack(idxs))
    assert torch.equal(bi, si) and torch.equal(bc, sc)


def _lut_rank(rank, world, port, path, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
    import torch
    import torch.distributed as dist
    import spart_amd
    from spart_amd import workloads
    torch.cuda.set_device(0)                                        # the ranks share the one GPU
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P = workloads.lhs_params(B, "full", seed=21)
    blk = spart_amd.generate_lut(P, "Sentinel2A-MSI", path=path, chunk=4096, shard=True)      # several chunks per rank
    mem = spart_amd.generate_lut(P, "Sentinel2A-MSI", chunk=4096, shard=True)                 # in memory: this rank's block only
    lo, hi = blk.rows
    ok_mem = mem.rows == (lo, hi) and all(mem[k].shape == (hi - lo, 13) and np.array_equal(mem[k], np.asarray(blk[k][lo:hi]))
                                          for k in ("R_TOC", "R_TOA", "L_TOA"))
    # observations: noisy copies of LUT rows + exact copies (ties cannot occur in an LHS table: add a duplicate row pair by
    # searching TWO stacked copies of the column below)
    meta, _, cols = spart_amd.load_lut(path)
    lut = np.asarray(cols["R_TOC"])
    rng = np.random.default_rng(4)
    obs = (lut[rng.integers(0, B, 300)] * (1 + 0.02 * rng.standard_normal((300, 13)))).astype(np.float32)
    obs[:5] = lut[[0, B - 1, B // 2, 7, B // 3]]
    idx, cost, st = spart_amd.invert_lut(path, obs, shard=True, stats=True)
    w = np.linspace(0.5, 2.0, 13).astype(np.float32)
    idx_w, cost_w = spart_amd.invert_lut(path, obs, weights=w, shard=True)
    twice = np.concatenate([lut, lut])                              # every row twice: the copy in the lower half must win
    idx_t, cost_t = spart_amd.invert_lut(twice, obs, shard=True)
    q.put((rank, (lo, hi), ok_mem, idx, cost, idx_w, cost_w, idx_t, cost_t, st["rows"], meta["rows"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_lut_rows_sharded_over_gloo_ranks_match_single_rank(world, tmp_path):
    """SURVEY 8(e) x (f3)/(f4): generate_lut(shard=True) -- every rank evaluates and writes its own contiguous block of the
    directory's .npy files, no gather -- and invert_lut(shard=True) -- local exact argmin + ONE all_gather of (cost, global row)
    -- with 2 and 3 ranks (ragged blocks) on the one GPU under gloo: the directory is byte-identical to a single-process
    run's, winners and costs equal the single-process search (and a brute force of the defined cost), lowest row on ties."""
    import torch.multiprocessing as mp
    import spart_amd
    from spart_amd import workloads
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from lut_brute_force import brute_force_numpy
    B = 30_001
    P = workloads.lhs_params(B, "full", seed=21)
    single = str(tmp_path / "single")
    spart_amd.generate_lut(P, "Sentinel2A-MSI", path=single, chunk=4096)
    path = str(tmp_path / "sharded")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_lut_rank, args=(r, world, port, path, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=600) for _ in range(world)), key=lambda g: g[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    per = -(-B // world)
    assert [g[1] for g in got] == [(min(B, r * per), min(B, (r + 1) * per)) for r in range(world)]
    assert all(g[2] for g in got) and all(g[9] == g[1] and g[10] == B for g in got)
    for f in ("R_TOC.npy", "R_TOA.npy", "L_TOA.npy", "params.npy", "meta.json"):
        assert open(os.path.join(path, f), "rb").read() == open(os.path.join(single, f), "rb").read(), f
    # inversion: the same answer on every rank, equal to the single-process call and to the brute force
    _, _, cols = spart_amd.load_lut(single)
    lut = np.asarray(cols["R_TOC"])
    obs_idx, obs_cost = got[0][3], got[0][4]
    for g in got[1:]:
        for a, b in zip(g[3:9], got[0][3:9]):
            assert np.array_equal(a, b)
    rng = np.random.default_rng(4)
    obs = (lut[rng.integers(0, B, 300)] * (1 + 0.02 * rng.standard_normal((300, 13)))).astype(np.float32)
    obs[:5] = lut[[0, B - 1, B // 2, 7, B // 3]]
    si, sc = spart_amd.invert_lut(single, obs)
    assert np.array_equal(obs_idx, si) and np.array_equal(obs_cost, sc)
    bi, bc = brute_force_numpy(lut, obs)
    assert np.array_equal(obs_idx, bi) and np.array_equal(obs_cost, bc)
    assert obs_idx[:5].tolist() == [0, B - 1, B // 2, 7, B // 3] and (obs_cost[:5] == 0).all()
    w = np.linspace(0.5, 2.0, 13).astype(np.float32)
    bi, bc = brute_force_numpy(lut, obs, w)
    assert np.array_equal(got[0][5], bi) and np.array_equal(got[0][6], bc)
    assert np.array_equal(got[0][7], obs_idx) and np.array_equal(got[0][8], obs_cost)      # duplicates: the lower copy wins


def test_bench_under_the_drivers_multi_rank_invocation():
    """The driver's exact N > 1 command line -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W` -- as a fresh child process, with two ranks sharing
    the one GPU of the test box under gloo (SPART_BENCH_BACKEND=gloo; the driver's run uses the default nccl = RCCL on N
    GPUs, which nothing here can measure): one JSON line from rank 0, strong scaling of ONE global table cut into contiguous
    shards, finite, and the gathered columns bit-identical to the single-rank run of the same batch (order-independent
    checksum of their bit patterns: nothing lost, nothing counted twice)."""
    import json
    import subprocess
    env = dict(os.environ, SPART_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--batch", "200000", "--steps", "3", "--warmup", "1", "--cpu-rows", "0"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", *common, "--no-extras"], env=env, capture_output=True,
                         text=True, timeout=600, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    l1 = [json.loads(l) for l in one.stdout.splitlines() if l.startswith("{")]
    assert len(l1) == 1
    a = l1[0]
    assert "multi_rank" not in a
    for world, rows in ((2, [100000, 100000]), (3, [66667, 66667, 66666])):      # (3 ranks: ragged shards, zero-padded gather block)
        two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
                              "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), *common],
                             env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert two.returncode == 0, two.stderr[-2000:]
        l2 = [json.loads(l) for l in two.stdout.splitlines() if l.startswith("{")]
        assert len(l2) == 1                                            # rank 0 prints ONE line
        b = l2[0]
        assert b["n_gpus"] == world and b["scaling"] == "strong" and b["steps"] == 3 and b["warmup"] == 1
        assert b["config"]["global_batch"] == 200000 and b["config"]["batch_per_gpu"] == rows[0] and b["config"]["finite"] is True
        assert b["metric"] == a["metric"] and b["unit"] == "spectra/s" and b["config"]["build_id"] == a["config"]["build_id"]
        assert b["config"]["columns_checksum"] == a["config"]["columns_checksum"]       # the gathered columns ARE the single-rank columns
        # (gloo moves the result blocks through host memory and the ranks share one GPU: the rate is a sanity bound only)
        assert 0.03 * a["value"] < b["value"] < 1.5 * a["value"], (a["value"], b["value"])
        # the self-diagnosis of a multi-rank run (nobody can watch the driver's 8-GPU job): every rank reported
        m = b["multi_rank"]
        assert m["ranks_seen"] == world and m["ranks"] == list(range(world)) and m["rows_per_rank"] == rows
        for k in ("per_rank_ms", "per_rank_compute_ms", "per_rank_band_kernel_ms", "per_rank_gather_ms"):
            assert len(m[k]) == world and all(np.isfinite(v) and v > 0 for v in m[k]), (k, m[k])
        assert m["gather_ms"] > 0 and np.isfinite(m["predicted_value"]) and m["predicted_value"] > 0
        assert max(m["per_rank_ms"]) <= b["ms_per_step"] * 1.001          # ms_per_step IS the maximum over the ranks' own clocks
        assert abs(m["gather_exposed_ms"] - (b["ms_per_step"] - max(m["per_rank_compute_ms"]))) < 1e-9
        # the LUT rows sharded over the same ranks (SURVEY 8e x 8f): generation without any collective, inversion with one
        # all_gather of the per-rank winners -- every rank took part, and rank 0 re-did the search on one device
        li, lg = b["configs"]["lut_invert"], b["configs"]["lut_generate"]
        assert li["ranks_seen"] == world and li["rows_per_rank"] == rows and len(li["per_rank_ms"]) == world
        assert li["winners_equal_to_single_device_search"] == li["checked"] == li["costs_bit_equal"] == 65536
        assert lg["ranks_seen"] == world and sum(lg["rows_per_rank"]) == 8 * 200000 and lg["finite_on_every_rank"] is True
        assert lg["value"] > 0 and li["value"] > 0
