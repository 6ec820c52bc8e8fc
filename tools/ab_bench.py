"""A/B timing of libspart_hip.so builds in ONE process, interleaved rounds (cdna guide rule 24).

    python tools/ab_bench.py name1=path1.so name2=path2.so [--batch 1000000] [--rounds 5] [--dtype float32]

Prints, per build, min / median milliseconds of the band kernel (HIP events inside the library) and of
the whole step (wall, synchronised), plus the max relative deviation of R_TOC/R_TOA/L_TOA from the
first build on the same inputs (sanity, not parity -- parity is tests/test_gpu_parity.py)."""
import argparse
import os
import statistics
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("builds", nargs="+")
    ap.add_argument("--batch", type=int, default=1_000_000)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--sensor", default="Sentinel2A-MSI")
    ap.add_argument("--prune", action="store_true")
    a = ap.parse_args()
    import torch
    from spart_amd import workloads
    from spart_amd.engine import Engine
    P = torch.as_tensor(workloads.lhs_params(a.batch, "full").T.copy(), device="cuda:0")
    engs = {}
    for b in a.builds:
        name, path = b.split("=", 1)
        engs[name] = Engine(a.sensor, 0, lib_path=path)
    ref = None
    band = {n: [] for n in engs}
    wall = {n: [] for n in engs}
    pre = {n: [] for n in engs}
    dev = {}
    for n, e in engs.items():
        o = e.run(P, a.dtype, prune=a.prune)
        torch.cuda.synchronize()
        cur = torch.stack([o[k].double() for k in ("R_TOC", "R_TOA", "L_TOA")])
        if ref is None:
            ref = cur
        dev[n] = ((cur - ref).abs() / ref.abs().clamp_min(1e-3)).max().item()
    import random
    random.seed(1)
    for _ in range(a.rounds):
        order = list(engs.items())
        random.shuffle(order)                     # (position in the round matters: clocks / thermal state)
        for n, e in order:
            e.profile(1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e.run(P, a.dtype, prune=a.prune)
            torch.cuda.synchronize()
            wall[n].append((time.perf_counter() - t0) * 1e3)
            st, _ = e.profile_read_stages()
            band[n].append(st["bands"])
            pre[n].append(st["prelude"])
            e.profile(0)
    # the column kernels' own durations: a second context per build with every kernel on one stream
    os.environ["SPART_SIDE_STREAM"] = "0"
    ser = {}
    for b in a.builds:
        name, path = b.split("=", 1)
        e = Engine(a.sensor, 0, lib_path=path)
        e.run(P, a.dtype, prune=a.prune)
        e.profile(5)
        for _ in range(5):
            e.run(P, a.dtype, prune=a.prune)
        st, k = e.profile_read_stages()
        e.profile(0)
        ser[name] = {q: v / k for q, v in st.items()}
    for n in engs:
        print(f"{n:24s} serial columns {ser[n]['columns']:.3f} | prelude min {min(pre[n]):6.3f} | band min {min(band[n]):8.3f} med {statistics.median(band[n]):8.3f} ms | "
              f"step min {min(wall[n]):8.3f} med {statistics.median(wall[n]):8.3f} ms | dev vs first {dev[n]:.2e}", flush=True)


if __name__ == "__main__":
    main()
