"""Headline benchmark: full SPART spectra/sec (R_TOC + R_TOA + L_TOA) at batch 1M per GPU.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path (spart_run_batch: prelude + fused PROSPECT/BSM/SAILH band
kernel over all 2162 bands + SMAC/TOC->TOA) over one batch of synthetic parameters already
resident in HBM, plus -- for N > 1 -- the single RCCL gather of the (B, nb, 3) result shards to
rank 0 (BASELINE.json north_star).  Workload = BASELINE config 4's generator (22-D Latin
hypercube, Sentinel2A-MSI, fp32 bands / fp64 sample scalars) at B = 1,000,000 per GPU
(weak scaling: every rank evaluates its own 1M-spectrum shard).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))

PROFILE_TAG = "r1_k"      # profiles/<tag>_traffic.json, <tag>_valu.json: the committed rocprofv3 PMC passes of this build
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_VALU_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 vector


def algorithmic_bytes(nb, dtype):
    """inputs + requested outputs per spectrum (inputs are always float64 here)."""
    es = 4 if dtype == "float32" else 8
    return 27 * 8 + 3 * nb * es


def measured_traffic(kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/<PROFILE_TAG>_traffic.json; FETCH_SIZE and WRITE_SIZE need separate passes, so this cannot be
    collected live).  None when the profile does not cover this kernel / batch."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", PROFILE_TAG + "_traffic.json")))
        for k, v in d.items():
            if k.replace(" ", "") == "spart::" + kernel.replace(" ", ""):
                return v["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def _cpu_worker(job):
    """One process of the CPU baseline: the oracle over its slice, 256-row blocks."""
    sensor, rows, seed, lo, hi, mode = job
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import spart_oracle as O
    from spart_amd import workloads
    T = O.load_tables()
    P = workloads.lhs_params(rows, "full", seed=seed)[lo:hi]
    kw = dict(e1="quad", pso="quad") if mode == "quad" else dict(pso="gl")
    if mode != "quad":
        O.spart_run(P[:8], sensor, T, **kw)              # warm-up (imports, table derivation)
    t0 = time.perf_counter()
    for i in range(0, len(P), 256):
        O.spart_run(P[i:i + 256], sensor, T, **kw)
    return time.perf_counter() - t0


def measured_valu(kern_s):
    """VALU issue figures of the band kernel from the committed SQ counter pass (profiles/<PROFILE_TAG>_valu.json):
    wave-instructions per launch, and the fraction of the chip's VALU issue slots they fill at the 2-cycle wave64
    fp32 cadence (1024 SIMDs, 2.4 GHz peak clock) over the launch duration measured in THIS run."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", PROFILE_TAG + "_valu.json")))
        n, t = d["SQ_INSTS_VALU"], d.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
        return {"wave_insts_per_launch": n, "transcendental_wave_insts": t,
                "issue_frac": n * 2.0 / (1024 * 2.4e9 * kern_s)}
    except Exception:
        return {}


def cpu_baseline(sensor, rows_per_core, seed):
    """The oracle (numpy port of the reference) timed on this box's host cores, bounded sample.  Runs BEFORE the
    GPU is initialised (worker processes are forked) and never touches it."""
    import multiprocessing as mp
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 16))                       # a one-GPU box gives this job a 16-core share
    rows = rows_per_core * cores
    ctx = mp.get_context("fork")
    bounds = [(i * rows // cores, (i + 1) * rows // cores) for i in range(cores)]
    with ctx.Pool(cores) as pool:
        t0 = time.perf_counter()
        busy = pool.map(_cpu_worker, [(sensor, rows, seed, lo, hi, "fast") for lo, hi in bounds])
        dt = time.perf_counter() - t0
        # the reference's own numerical route (scipy quad for E1 and the 61 hot-spot integrals), 4 rows per core
        q0 = time.perf_counter()
        pool.map(_cpu_worker, [(sensor, 4 * cores, seed, 4 * i, 4 * i + 4, "quad") for i in range(cores)])
        qdt = time.perf_counter() - q0
    return {"value": rows / dt, "unit": "spectra/s", "cores": cores, "kind": "port",
            "sample": f"{rows} rows of the same LHS workload, oracle/spart_oracle.py (vectorised numpy, closed-form E1 / "
                      f"Gauss-Legendre hot spot), {cores} processes, {dt:.1f} s wall ({sum(busy):.0f} s CPU)",
            "per_core": rows / sum(busy),
            "reference_route": {"value": 4 * cores / qdt, "unit": "spectra/s", "cores": cores,
                                "sample": f"{4 * cores} rows, same oracle with the reference's scipy-quad E1 and hot-spot "
                                          f"integrals (e1='quad', pso='quad'), {qdt:.1f} s wall"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1_000_000, help="spectra per GPU per step")
    ap.add_argument("--dtype", default="float32", choices=["float32", "float64"])
    ap.add_argument("--sensor", default="Sentinel2A-MSI")
    ap.add_argument("--cpu-rows", type=int, default=8192, help="rows PER HOST CORE for the CPU baseline (0 = skip)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    cpu = None
    if world == 1 and args.cpu_rows > 0:
        from spart_amd import workloads as _w            # (no torch / HIP import yet: the workers are forked)
        cpu = cpu_baseline(args.sensor, args.cpu_rows, _w.LHS_SEED)

    import numpy as np
    import torch
    import torch.distributed as dist
    from spart_amd import get_engine, workloads

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    # SPART_BENCH_BACKEND=gloo + several ranks on one GPU is a rehearsal mode for the 1-GPU box (it exercises
    # the shard / double-buffer / gather logic); the driver's multi-GPU run uses the default "nccl" = RCCL.
    backend = os.environ.get("SPART_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    B = args.batch
    eng = get_engine(args.sensor, dev_index)
    nb = eng.nb
    # synthetic inputs: this rank's shard of the LHS workload, resident in HBM before timing starts
    P = torch.as_tensor(workloads.lhs_params(B, "full", seed=workloads.LHS_SEED + rank).T.copy(), device=dev)
    td = torch.float32 if args.dtype == "float32" else torch.float64
    # (3, B, nb) so that the three result columns travel in ONE gather.  Two result buffers: the gather of
    # step i (RCCL's own stream) overlaps the kernels of step i + 1 (compute stream); a buffer is reused
    # only after its gather has completed.  Every gather is inside the timed region (fence() waits for all).
    nbuf = 2 if world > 1 else 1
    res = [torch.empty((3, B, nb), dtype=td, device=dev) for _ in range(nbuf)]
    outs = [{"R_TOC": r[0], "R_TOA": r[1], "L_TOA": r[2]} for r in res]
    gather_lists = [[torch.empty_like(res[0]) for _ in range(world)] if rank == 0 else None
                    for _ in range(nbuf)] if world > 1 else None
    works = [None] * nbuf
    counter = [0]

    def step():
        j = counter[0] % nbuf
        counter[0] += 1
        if works[j] is not None:
            works[j].wait()                     # compute stream waits until buffer j's previous gather is done
            works[j] = None
        eng.run(P, args.dtype, out=outs[j])     # opt = NULL: all 2162 bands of every spectrum are evaluated
        if world > 1:
            works[j] = dist.gather(res[j], gather_lists[j], dst=0, async_op=True)

    def fence():
        for j in range(nbuf):
            if works[j] is not None:
                works[j].wait()
                works[j] = None
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    eng.profile(args.steps)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    band_ms, ncalls = eng.profile_read()
    eng.profile(0)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ok = all(bool(torch.isfinite(r).all().item()) for r in res)
        total = B * world * args.steps
        value = total / dt
        kern_s = band_ms / max(ncalls, 1) / 1e3
        abytes = algorithmic_bytes(nb, args.dtype) * B          # per launch of the band kernel's step
        std = args.dtype == "float32" and B == 1_000_000 and args.sensor == "Sentinel2A-MSI"   # the profiled configuration
        achieved = abytes / kern_s / 1e9
        line = {
            "metric": "SPART spectra/sec (R_TOC+R_TOA+L_TOA) at batch 1M; achieved HBM GB/s vs peak",
            "value": value, "unit": "spectra/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.dtype == "float32" else "f64", "data": "synthetic",
            "config": {"workload": "full SPART (BSM+PROSPECT-5D+SAILH+SMAC), 22-D Latin hypercube (seed 20240613+rank), "
                                   f"{args.sensor}, all 2162 bands evaluated per spectrum, columns-only output",
                       "batch_per_gpu": B, "global_batch": B * world, "bands_evaluated": 2162, "sensor_bands": nb,
                       "parallelism": f"dp{world} (independent shards + one RCCL gather to rank 0 per step, overlapped with "
                                      "the next step's kernels)" if world > 1 else "single GPU",
                       "input_dtype": "f64", "finite": ok},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": measured_traffic("k_bands<float,0,1>") if std else None,
                         "kernel": "k_bands<float,0,1>" if args.dtype == "float32" else "k_bands<double,0,1>",
                         "kernel_ms": kern_s * 1e3, "algorithmic_bytes_per_spectrum": algorithmic_bytes(nb, args.dtype),
                         "note": "fused path is VALU/transcendental bound by design (SURVEY.md §8d); HBM fraction is "
                                 "reported because the metric asks for it",
                         "valu": dict({"flop_eq_per_spectrum": 8.7e5,
                                       "achieved_tflop_eq": 8.7e5 * B / kern_s / 1e12, "peak_fp32_tflops": FP32_VALU_TFLOPS},
                                      **(measured_valu(kern_s) if std else {}))},
        }
        line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
