"""Copy the summaries of gpurun_out/<tag>/ (tools/collect_profiles.sh) into profiles/<tag>_* and derive <tag>_traffic.json."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
res = {}
for d, c in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv"))[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c and "spart" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res.setdefault(k, {})[c + "_KB_per_launch"] = sum(v) / len(v)
    shutil.copy(f, os.path.join(dst, f"{tag}_{d}_counter_collection.csv"))
for k, v in res.items():
    v["hbm_bytes_per_launch"] = (v.get("FETCH_SIZE_KB_per_launch", 0) + v.get("WRITE_SIZE_KB_per_launch", 0)) * 1024
res["_note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 3 --warmup 1 --cpu-rows 0` "
                "(B = 1M, Sentinel2A, fp32); KB per launch as rocprofv3 reports them, hbm_bytes = (FETCH+WRITE)*1024. Reads are "
                "4-B-per-lane coalesced loads of the 192 B/sample constants and writes are 16-B / 4-B per-lane stores, for which "
                "the gfx950 FETCH_SIZE 1/2-factor of wide (16 B/lane) streams is not calibrated (MI355X_MICROARCH.md, HBM); no "
                "correction applied.")
json.dump(res, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
shutil.copy(glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
shutil.copy(glob.glob(os.path.join(src, "pmc_sq", "*", "*_counter_collection.csv"))[0], os.path.join(dst, f"{tag}_pmc_sq_counter_collection.csv"))
# per-launch SQ counters of the band kernel -> <tag>_valu.json (read by bench.py for roofline.valu)
sq = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(os.path.join(dst, f"{tag}_pmc_sq_counter_collection.csv"))):
    if "k_bands" in r["Kernel_Name"]:
        sq[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
if sq:
    names = sorted(next(iter(sq.values())))
    avg = {n: sum(d[n] for d in sq.values()) / len(sq) for n in names}
    avg["_note"] = ("per launch of the band kernel (B = 1M, fp32), summed over all XCDs / SEs; GRBM_GUI_ACTIVE is the sum over the 8 XCDs "
                    "(divide by 8 for the kernel's cycles)")
    json.dump(avg, open(os.path.join(dst, f"{tag}_valu.json"), "w"), indent=1)
for d in ("mat_stats", "mat_pmc_write"):
    g = glob.glob(os.path.join(src, d, "*", "*_kernel_stats.csv" if d == "mat_stats" else "*_counter_collection.csv"))
    if g:
        shutil.copy(g[0], os.path.join(dst, f"{tag}_{d}.csv"))
if os.path.exists(os.path.join(src, "mat_bench.txt")):
    shutil.copy(os.path.join(src, "mat_bench.txt"), os.path.join(dst, f"{tag}_mat_bench.txt"))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{tag}_bench.json"))
shutil.copy(os.path.join(src, "configs.json"), os.path.join(dst, f"{tag}_configs.json"))
print(json.dumps({k: v for k, v in res.items() if k != "_note"}, indent=1))
print(open(os.path.join(src, "bench.json")).read()[:300])
