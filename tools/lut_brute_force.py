"""Brute-force checkers of spart_lut_nearest (tests/ and bench.py; tooling, not product).

The cost is DEFINED (include/spart_hip.h) as the sequential evaluation, in the call's dtype and without fused multiply-adds,

    c = 0;  for j = 0 .. nb-1:  d = lut[b, j] - obs[m, j];  c = c + (w_j * d) * d        (w = None: c = c + d * d)

and the answer as the LOWEST row index attaining the minimum; rows whose cost is NaN / +inf never win (-1 / +inf).  This is the
reference's own nearest-index rule -- an exact np.argmin, first index on ties (/root/reference/src/SPART/SPART.py:381-387) --
applied to the weighted squared distance.  numpy / eager torch evaluate every elementwise operation on its own, rounded to the
array dtype, so the loops below ARE that definition.
"""
import numpy as np


def brute_force_numpy(lut, obs, w=None):
    """lut (B, nb), obs (M, nb), w (nb,) or None, all of ONE float dtype -> (idx (M,) int64, cost (M,) dtype)"""
    lut = np.ascontiguousarray(lut)
    dt = lut.dtype
    obs = np.ascontiguousarray(obs, dtype=dt)
    w = None if w is None else np.asarray(w, dtype=dt)
    B, nb = lut.shape
    M = obs.shape[0]
    idx = np.full(M, -1, dtype=np.int64)
    cost = np.full(M, np.inf, dtype=dt)
    cols = [np.ascontiguousarray(lut[:, j]) for j in range(nb)]
    with np.errstate(all="ignore"):
        for m in range(M):
            c = np.zeros(B, dtype=dt)
            for j in range(nb):
                d = cols[j] - obs[m, j]
                t = d if w is None else w[j] * d
                c = c + t * d
            c[~np.isfinite(c)] = np.inf
            i = int(np.argmin(c))                     # first index of the minimum
            if c[i] < np.inf:
                idx[m], cost[m] = i, c[i]
    return idx, cost


def brute_force_torch(lut, obs, w=None, max_elems=1 << 27):
    """the same on the GPU with eager torch ops (one kernel per operation: no contraction), in blocks of observations.
    lut (B, nb), obs (M, nb), w (nb,) or None: tensors of one dtype on one device -> (idx int64, cost)"""
    import torch
    B, nb = lut.shape
    M = obs.shape[0]
    inf = float("inf")
    cols = [lut[:, j].contiguous() for j in range(nb)]
    rows = torch.arange(B, device=lut.device, dtype=torch.int64)
    idx = torch.full((M,), -1, dtype=torch.int64, device=lut.device)
    cost = torch.full((M,), inf, dtype=lut.dtype, device=lut.device)
    mb = max(1, min(M, max_elems // max(B, 1)))
    for m0 in range(0, M, mb):
        o = obs[m0:m0 + mb]
        c = torch.zeros((o.shape[0], B), dtype=lut.dtype, device=lut.device)
        for j in range(nb):
            d = cols[j][None, :] - o[:, j][:, None]
            t = d if w is None else w[j] * d
            c = c + t * d
        c = torch.where(torch.isfinite(c), c, torch.full_like(c, inf))
        cmin = c.min(dim=1).values
        first = torch.where(c == cmin[:, None], rows[None, :], torch.full_like(rows, B)[None, :]).min(dim=1).values
        ok = cmin < inf
        idx[m0:m0 + mb] = torch.where(ok, first, torch.full_like(first, -1))
        cost[m0:m0 + mb] = cmin
    return idx, cost
