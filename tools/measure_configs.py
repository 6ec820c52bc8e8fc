"""Measure the BASELINE.json configurations 2-5 on one MI355X (everything that fits one GPU) and print one
JSON object: throughput, algorithmic-byte rates, fp32-vs-fp64-vs-oracle tolerance table.

    python tools/measure_configs.py [--quick]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def timeit(fn, torch, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return min(ts), sorted(ts)[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    import numpy as np
    import torch
    import spart_oracle as O
    from spart_amd import get_engine, workloads

    out = {}
    dev = "cuda:0"
    T = O.load_tables()

    # ---- config 2: PROSPECT-5D leaf only, 10k x 2001, fp64
    eng0 = get_engine(None, 0)
    P = workloads.lhs_params(10_000, "leaf")
    cols = [torch.as_tensor(P[:, i].copy(), device=dev) for i in range(9)]
    tmin, tmed = timeit(lambda: eng0.prospect(cols, "float64"), torch)
    r, t, k = eng0.prospect(cols, "float64")
    ro, to, ko = O.prospect_5d(P[:512, :9], T)
    err = max(float(np.max(np.abs(r[:512].cpu().numpy() - ro) / np.maximum(np.abs(ro), 0.1))),
              float(np.max(np.abs(t[:512].cpu().numpy() - to) / np.maximum(np.abs(to), 0.1))))
    out["config2_prospect_10k_fp64"] = {"leaf_spectra_per_s": 10_000 / tmin, "ms": tmin * 1e3,
                                        "algorithmic_GBps": 48096 * 10_000 / tmin / 1e9,
                                        "hbm_frac_of_8TBps": 48096 * 10_000 / tmin / 8e12,
                                        "max_rel_err_vs_oracle_floor0.1": err}
    Pl = workloads.lhs_params(1_000_000 if not a.quick else 100_000, "leaf")
    cols = [torch.as_tensor(Pl[:, i].copy(), device=dev) for i in range(9)]
    n = Pl.shape[0]
    for dt, es in (("float64", 8), ("float32", 4)):
        tmin, _ = timeit(lambda: eng0.prospect(cols, dt), torch, n=3, warm=1)
        out[f"prospect_{n}_{dt}"] = {"leaf_spectra_per_s": n / tmin, "ms": tmin * 1e3,
                                     "algorithmic_GBps": (72 + 3 * 2001 * es) * n / tmin / 1e9}
    del cols
    torch.cuda.empty_cache()

    # ---- config 3: full SPART, 100k, S2A, fp32
    eng = get_engine("Sentinel2A-MSI", 0)
    P3 = torch.as_tensor(workloads.lhs_params(100_000, "full").T.copy(), device=dev)
    tmin, _ = timeit(lambda: eng.run(P3, "float32"), torch)
    out["config3_full_100k_S2A_fp32"] = {"spectra_per_s": 100_000 / tmin, "ms": tmin * 1e3}

    # ---- config 4 shape on one GPU: 1M, fp32 and fp64; pruned; materialised
    B = 1_000_000 if not a.quick else 200_000
    P4 = torch.as_tensor(workloads.lhs_params(B, "full").T.copy(), device=dev)
    for dt in ("float32", "float64"):
        tmin, _ = timeit(lambda: eng.run(P4, dt), torch, n=3, warm=1)
        out[f"config4_full_{B}_S2A_{dt}"] = {"spectra_per_s": B / tmin, "ms": tmin * 1e3}
    tmin, _ = timeit(lambda: eng.run(P4, "float32", prune=True), torch, n=3, warm=1)
    out[f"pruned_{B}_S2A_float32"] = {"spectra_per_s": B / tmin, "ms": tmin * 1e3,
                                     "note": "prune_unused_bands=1: NOT full spectra, reported separately"}
    Bm = 200_000
    Pm = P4[:, :Bm].contiguous()
    fields = ("leaf_refl", "leaf_tran", "leaf_kchl", "soil_refl", "soil_refl_dry", "rso", "rdo", "rsd", "rdd")
    nbytes = (7 * 2162 + 2 * 2001) * 4 * Bm
    holder = {}

    def mat():
        holder["o"] = eng.run(Pm, "float32", materialize=fields)
    tmin, _ = timeit(mat, torch, n=3, warm=1)
    out[f"materialised_{Bm}_S2A_float32"] = {"spectra_per_s": Bm / tmin, "ms": tmin * 1e3,
                                            "written_GBps": nbytes / tmin / 1e9, "hbm_frac_of_8TBps": nbytes / tmin / 8e12}
    holder.clear()
    torch.cuda.empty_cache()

    # ---- configs 4 / 5: fp32 vs fp64 tolerance sweep over the whole 1M batch, both vs the oracle on 2048 rows
    def sweep(kind, sensor):
        e_ = get_engine(sensor, 0)
        Ph = workloads.lhs_params(B, kind)
        Pd = torch.as_tensor(Ph.T.copy(), device=dev)
        o64 = {k: v.clone() for k, v in e_.run(Pd, "float64").items()}
        o32 = e_.run(Pd, "float32")
        tab = {}
        for k in ("R_TOC", "R_TOA", "L_TOA"):
            d = (o32[k].double() - o64[k]).abs()
            ref = o64[k].abs()
            r6 = d / ref.clamp_min(1e-6)
            r3 = d / ref.clamp_min(1e-3)
            samp = r3.flatten()[torch.randperm(r3.numel(), device=dev)[:4_000_000]]
            tab[k] = {"fp32_vs_fp64": {
                "max_abs_err": float(d.max()),
                "max_rel_floor1e-6": float(r6.max()), "max_rel_floor1e-3": float(r3.max()),
                "median_rel": float(samp.median()), "p99_rel": float(torch.quantile(samp, 0.99)),
                "p99.9_rel": float(torch.quantile(samp, 0.999)),
                "frac_entries_rel>1e-4_floor1e-3": float((r3 > 1e-4).double().mean()),
                "frac_entries_rel>1e-4_floor1e-6": float((r6 > 1e-4).double().mean())}}
        nref = 2048 if not a.quick else 256
        refo = O.spart_run(Ph[:nref], sensor, T, pso="gl")
        for k in ("R_TOC", "R_TOA", "L_TOA"):
            for name, o in (("fp64", o64), ("fp32", o32)):
                g = o[k][:nref].double().cpu().numpy()
                tab[k][f"{name}_vs_oracle_{nref}rows"] = {
                    "max_rel_floor1e-6": float(np.max(np.abs(g - refo[k]) / np.maximum(np.abs(refo[k]), 1e-6))),
                    "max_rel_floor1e-3": float(np.max(np.abs(g - refo[k]) / np.maximum(np.abs(refo[k]), 1e-3)))}
        for dt in ("float32", "float64"):
            tmin, _ = timeit(lambda: e_.run(Pd, dt), torch, n=3, warm=1)
            tab[f"spectra_per_s_{dt}"] = B / tmin
        return tab
    out[f"config4_full_{B}_S2A_tolerance"] = sweep("full", "Sentinel2A-MSI")
    out[f"config5_pro_{B}_S2B_tolerance"] = sweep("pro", "Sentinel2B-MSI")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
