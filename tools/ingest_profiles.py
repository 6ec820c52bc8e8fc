"""Copy the summaries of gpurun_out/<tag>/ (tools/collect_profiles.sh) into profiles/<tag>_* and derive the
per-kernel counter files bench.py reads: profiles/<round>_counters_<dtype>.json.

    python tools/ingest_profiles.py r2_a [r2]     (second argument: the prefix bench.py's PROFILE_TAG names)"""
import collections
import csv
import glob
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
tag = sys.argv[1]
prefix = sys.argv[2] if len(sys.argv) > 2 else tag.split("_")[0]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")


def src_hash():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "spart-python_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:12]


def one(pattern):
    g = glob.glob(os.path.join(src, pattern))
    return g[0] if g else None


def per_kernel(f, counters=None):
    """{kernel: {counter: average per dispatch}} (a counter's rows of one dispatch are summed: XCDs / SEs)"""
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for r in csv.DictReader(open(f)):
        if counters and r["Counter_Name"] not in counters:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("spart::", "")
        acc[k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: {c: sum(d.values()) / len(d) for c, d in v.items()} for k, v in acc.items()}


# ---- calibration of FETCH_SIZE / WRITE_SIZE on 1 GiB moved in this library's access widths
calib = {}
for d, c in (("calib_fetch", "FETCH_SIZE"), ("calib_write", "WRITE_SIZE")):
    f = one(f"{d}/*/*_counter_collection.csv")
    if f:
        shutil.copy(f, os.path.join(dst, f"{tag}_{d}.csv"))
        for k, v in per_kernel(f, {c}).items():
            if (c == "FETCH_SIZE") == k.startswith("read"):
                calib[k] = v[c] * 1024 / float(1 << 30)          # counter bytes per byte actually moved
if calib:
    json.dump({"_note": "rocprofv3 counter (KB x 1024) / bytes actually moved (1 GiB per kernel, tools/ubench/fetch_calib.hip): "
                        "FETCH_SIZE for the read_* kernels, WRITE_SIZE for the write_* kernels", **calib},
              open(os.path.join(dst, f"{tag}_counter_calibration.json"), "w"), indent=1)


def corr(kind, width):
    """divide a counter by this to get bytes: calibration of the nearest access pattern, 1.0 when not calibrated"""
    name = {("r", 4): "read_coalesced<float>", ("r", 8): "read_coalesced<double>", ("r", 16): "read_coalesced<HIP_vector_type<float, 4u> >",
            ("r", "seg"): "read_segments", ("w", 4): "write_coalesced<float>", ("w", 8): "write_coalesced<double>",
            ("w", 16): "write_coalesced<HIP_vector_type<float, 4u> >", ("w", "g"): "write_strided16"}[(kind, width)]
    return calib.get(name, 1.0)


# ---- the step's kernels, per dtype
for dt in ("float32", "float64"):
    fs = {c: one(f"pmc_{d}_{dt}/*/*_counter_collection.csv") for d, c in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE"), ("sq", "SQ"))}
    if not all(fs.values()):
        continue
    kern = collections.defaultdict(dict)
    for c, f in fs.items():
        shutil.copy(f, os.path.join(dst, f"{tag}_pmc_{c.lower().replace('_size', '')}_{dt}.csv"))
        for k, v in per_kernel(f).items():
            if not k.startswith("k_"):
                continue
            for cn, val in v.items():
                kern[k][cn if cn not in ("FETCH_SIZE", "WRITE_SIZE") else cn + "_KB"] = val
    es = 4 if dt == "float32" else 8
    for k, v in kern.items():
        v["in_step"] = k != "k_econv"
        # dominant access width of each kernel's HBM streams (spart_kernels.h): the prelude reads 8 B/lane parameter
        # columns and writes 4 / 8 B/lane constant rows; the band kernel reads 128-B constant segments and writes
        # the 4-B / 8-B per-lane band sums; the column kernel reads 8 B/lane rows and writes dtype-sized columns
        if k.startswith("k_prelude"):
            fr, fw = corr("r", 8), corr("w", 8)
        elif k.startswith("k_bands"):
            fr, fw = (corr("r", "seg") if es == 4 else corr("r", 8)), corr("w", es)
        else:
            fr, fw = corr("r", 8), corr("w", es)
        v["fetch_bytes"] = v.get("FETCH_SIZE_KB", 0.0) * 1024 / fr
        v["write_bytes"] = v.get("WRITE_SIZE_KB", 0.0) * 1024 / fw
        v["hbm_bytes"] = v["fetch_bytes"] + v["write_bytes"]
        v["calibration"] = {"fetch": fr, "write": fw}
    out = {"src_hash": src_hash(), "batch": 1_000_000, "nb": 13, "sensor": "Sentinel2A-MSI", "dtype": dt,
           "command": f"rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE | SQ set> -- python3 bench.py --steps 3 --warmup 1 "
                      f"--dtype {dt} --cpu-rows 0 --no-extras  (three separate passes, tools/collect_profiles.sh {tag})",
           "correction": "FETCH_SIZE / WRITE_SIZE (KB x 1024) divided by the factor measured on 1 GiB moved with the same "
                         f"access width (profiles/{tag}_counter_calibration.json; MI355X_MICROARCH.md, HBM: gfx950 tallies wide "
                         "reads at 1/2); Infinity-Cache hits are counted as traffic",
           "kernels": kern}
    json.dump(out, open(os.path.join(dst, f"{prefix}_counters_{dt}.json"), "w"), indent=1)
    f = one(f"stats_{dt}/*/*_kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(dst, f"{tag}_kernel_stats_{dt}.csv"))
    print(dt, {k: (round(v["hbm_bytes"] / 1e6, 1), round(v.get("SQ_INSTS_VALU", 0) / 1e9, 3)) for k, v in kern.items()})

# ---- the modes beside the headline (tools/mode_run.py): one counter file each, read by bench.py's mode records
MODE_BATCH = {"materialized": 200_000, "pruned": 1_000_000, "lut_invert": 1_000_000}
for mode, batch in MODE_BATCH.items():
    kern = collections.defaultdict(dict)
    for d in ("fetch", "write", "sq"):
        f = one(f"{mode}_pmc_{d}/*/*_counter_collection.csv")
        if not f:
            continue
        shutil.copy(f, os.path.join(dst, f"{tag}_{mode}_pmc_{d}.csv"))
        for k, v in per_kernel(f).items():
            if k.startswith("k_"):
                for cn, val in v.items():
                    kern[k][cn if cn not in ("FETCH_SIZE", "WRITE_SIZE") else cn + "_KB"] = val
    f = one(f"{mode}_stats/*/*_kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(dst, f"{tag}_{mode}_kernel_stats.csv"))
        for r in csv.DictReader(open(f)):
            k = r["Name"].split("(")[0].replace("void ", "").replace("spart::", "")
            if k in kern:
                kern[k]["avg_ms"] = float(r["AverageNs"]) / 1e6
                kern[k]["calls"] = int(r["Calls"])
    if not kern:
        continue
    for k, v in kern.items():
        # materialised spectra: 4-byte coalesced stores (1.00), constants read as 128-byte segments; the LUT tiles are read as
        # coalesced 4-byte loads (gfx950 tallies them at 1/2) -- and mostly from L2 / Infinity Cache, which FETCH_SIZE counts too
        fr = corr("r", "seg") if k.startswith("k_bands") else corr("r", 4 if "lut" in k else 8)
        fw = corr("w", 4) if (k.startswith("k_bands") or "lut" in k) else corr("w", 8)
        v["fetch_bytes"] = v.get("FETCH_SIZE_KB", 0.0) * 1024 / fr
        v["write_bytes"] = v.get("WRITE_SIZE_KB", 0.0) * 1024 / fw
        v["hbm_bytes"] = v["fetch_bytes"] + v["write_bytes"]
        v["calibration"] = {"fetch": fr, "write": fw}
    json.dump({"src_hash": src_hash(), "batch": batch, "nb": 13, "mode": mode,
               "command": f"rocprofv3 --kernel-trace --stats | --pmc <FETCH_SIZE | WRITE_SIZE | SQ set> -- python3 tools/mode_run.py {mode} "
                          f"(separate passes, tools/collect_profiles.sh {tag})", "kernels": kern},
              open(os.path.join(dst, f"{prefix}_counters_{mode}.json"), "w"), indent=1)
    print(mode, {k: (round(v["hbm_bytes"] / 1e6, 1), round(v.get("avg_ms", 0), 3)) for k, v in kern.items()})

# ---- config 2 (PROSPECT-only kernel at 10k x 2001 float64)
c2 = {}
for d in ("fetch", "write", "sq"):
    f = one(f"c2_pmc_{d}/*/*_counter_collection.csv")
    if f:
        shutil.copy(f, os.path.join(dst, f"{tag}_c2_pmc_{d}.csv"))
        for k, v in per_kernel(f).items():
            if k.startswith("k_"):
                c2.setdefault(k, {}).update(v)
f = one("c2_stats/*/*_kernel_stats.csv")
if f:
    shutil.copy(f, os.path.join(dst, f"{tag}_c2_kernel_stats.csv"))
if c2:
    json.dump({"src_hash": src_hash(), "batch": 10_000, "kernels": c2,
               "_note": "per launch, BASELINE config 2 (tools/prospect_bench.py 10000 float64); FETCH_SIZE / WRITE_SIZE in KB as reported"},
              open(os.path.join(dst, f"{tag}_c2_counters.json"), "w"), indent=1)
for name in ("bench.json", "c2_bench.txt", "mode_cost.txt", "lut_rate.txt", "lut_invert_rate.txt", "mat_bench.txt", "fast_prelude_dev.txt", "power_materialized.txt",
             "power_headline.txt", "power_cap.txt", "parity_262144rows.json"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, f"{tag}_{name}"))
print(json.dumps(calib, indent=1))
