"""Multi-GPU evaluation: the batch is cut into contiguous shards, one per rank (one process per
GPU), every rank evaluates its shard independently (no data-path collective), and the
(3, b, nb) result blocks are collected on rank 0 with ONE gather (RCCL over xGMI when the
process group is 'nccl').  Nothing here touches HIP directly, so the same code
runs under the 'gloo' backend in the CPU tests with an injected evaluator.

The LUT rows of SURVEY.md §8(f) shard the same way (contiguous blocks of ``ceil(B / world)`` rows, rank order = row order):
* generation (spart_amd.lut.generate_lut(shard=True)): every rank evaluates its block and writes it into the directory's
  .npy files at its own rows -- no collective on the data path at all, rank 0 writes the manifest;
* inversion (lut_nearest_sharded below): every rank finds the exact nearest row of ITS block for every observation, then
  ONE all_gather of (cost, global row index) per observation -- 16 bytes x M per rank -- and the same selection rule on
  every rank: lowest cost, lowest row index on ties (the reference's np.argmin rule, SPART.py:381-387)."""


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


def _cost_bits(cost):
    """cost (M,) float32 / float64 -> int64 carrying its bit pattern (the collective moves ONE integer tensor)"""
    import torch
    if cost.dtype == torch.float64:
        return cost.contiguous().view(torch.int64)
    return cost.contiguous().view(torch.int32).to(torch.int64)


def _cost_from_bits(bits, dtype):
    import torch
    if dtype == torch.float64:
        return bits.contiguous().view(torch.float64)
    return bits.to(torch.int32).contiguous().view(torch.float32)


def select_nearest(costs, idxs):
    """The winner per observation among per-shard winners: costs (W, M) float, idxs (W, M) int64 GLOBAL row indices (-1 = the
    shard has no row with a finite cost, cost +inf).  Lowest cost, lowest row index on ties; (-1, +inf) when no shard has one.
    Pure tensor arithmetic on whatever device the inputs live on."""
    import torch
    big = torch.iinfo(torch.int64).max
    cmin = costs.min(dim=0).values
    cand = torch.where((costs == cmin.unsqueeze(0)) & (idxs >= 0), idxs, torch.full_like(idxs, big))
    best = cand.min(dim=0).values
    none = best == big
    best = torch.where(none, torch.full_like(best, -1), best)
    cost = torch.where(none, torch.full_like(cmin, float("inf")), cmin)
    return best, cost


def lut_nearest_sharded(lut_local, row0, obs, nearest, group=None, comm_device=None):
    """LUT inversion with the LUT ROW-SHARDED over the ranks of ``group``.

    lut_local : this rank's contiguous block of LUT rows, (b, nb) (b may be 0)
    row0      : global index of its first row (shard_bounds(B, world, rank)[0])
    obs       : (M, nb) observations, THE SAME on every rank
    nearest   : nearest(lut_local, obs) -> (idx (M,) int64 local row index or -1, cost (M,)): the exact single-device search
                (Engine.lut_nearest; the CPU tests inject a brute force)
    comm_device : where the (M, 2) winners live for the collective: None = where ``nearest`` returned them (device tensors under
                'nccl' = RCCL), "cpu" for a 'gloo' group
    Returns (idx (M,) int64 GLOBAL row index, cost (M,)) on EVERY rank -- bit-identical to the single-device search over
    the whole LUT: a row's cost does not depend on the other rows, so the per-shard minima are the same numbers and the
    lowest-index tie rule composes.  One all_gather of an (M, 2) int64 tensor (cost bits, global index) per call.
    (The single-device search excludes LUT rows whose CENTRED norm overflows the dtype; the centre is the block's, so for
    values near the largest finite number the excluded rows can differ from the unsharded call -- include/spart_hip.h.)"""
    import torch
    import torch.distributed as dist

    M = obs.shape[0]
    if lut_local.shape[0] > 0 and M > 0:
        idx, cost = nearest(lut_local, obs)
        idx = torch.where(idx >= 0, idx + int(row0), idx)
    else:
        cdt = obs.dtype if obs.dtype in (torch.float32, torch.float64) else torch.float32
        idx = torch.full((M,), -1, dtype=torch.int64, device=obs.device)
        cost = torch.full((M,), float("inf"), dtype=cdt, device=obs.device)
    if comm_device is not None:
        idx, cost = idx.to(comm_device), cost.to(comm_device)
    if not (dist.is_available() and dist.is_initialized()):
        return idx, cost
    world = dist.get_world_size(group)
    pack = torch.stack([_cost_bits(cost), idx.to(torch.int64)], dim=1).contiguous()         # (M, 2) int64
    bufs = [torch.empty_like(pack) for _ in range(world)]
    dist.all_gather(bufs, pack, group=group)
    allp = torch.stack(bufs)                                                                 # (W, M, 2)
    costs = torch.stack([_cost_from_bits(allp[r, :, 0], cost.dtype) for r in range(world)])
    return select_nearest(costs, allp[:, :, 1])
