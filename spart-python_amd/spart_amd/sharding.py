"""Multi-GPU evaluation: the batch is cut into contiguous shards, one per rank (one process per
GPU), every rank evaluates its shard independently (no data-path collective), and the
(3, b, nb) result blocks are collected on rank 0 with ONE gather (RCCL over xGMI when the
process group is 'nccl').  Nothing here touches HIP directly, so the same code
runs under the 'gloo' backend in the CPU tests with an injected evaluator."""
import numpy as np


def shard_bounds(B, world, rank):
    """[lo, hi) of rank's contiguous block of ceil(B / world) samples (the last ranks may be short or empty)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world / rank")
    per = -(-B // world)
    lo = min(B, rank * per)
    return lo, min(B, lo + per)


def default_evaluate(sensor, dtype, device=None):
    """evaluate(P_shard (27,b) tensor) -> (3, b, nb) tensor [R_TOC, R_TOA, L_TOA] on the HIP engine."""
    from .engine import get_engine

    def ev(P):
        import torch
        eng = get_engine(sensor, device)
        b = P.shape[1]
        td = torch.float32 if dtype in ("float32", "fp32", "f32") else torch.float64
        res = torch.empty((3, b, eng.nb), dtype=td, device=eng.device)
        if b:
            eng.run(P.to(eng.device), dtype, out={"R_TOC": res[0], "R_TOA": res[1], "L_TOA": res[2]})
        return res
    return ev


def run_sharded(P, evaluate, group=None, dst=0):
    """P: (27, B) tensor holding the WHOLE batch on every rank (parameters are tiny: 216 B / spectrum).

    Returns the (3, B, nb) result on rank ``dst`` and None elsewhere.  Without an initialised
    process group this is a plain single-device evaluation.
    """
    import torch
    import torch.distributed as dist

    B = P.shape[1]
    if not (dist.is_available() and dist.is_initialized()):
        return evaluate(P)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(B, world, rank)
    local = evaluate(P[:, lo:hi].contiguous())
    per = -(-B // world)
    nb = local.shape[2]
    if local.shape[1] != per:                       # short / empty tail shard: pad so that gather sizes match
        pad = torch.zeros((3, per, nb), dtype=local.dtype, device=local.device)
        pad[:, :local.shape[1]] = local
        local = pad
    local = local.contiguous()
    bufs = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    dist.gather(local, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat(bufs, dim=1)[:, :B].contiguous()
