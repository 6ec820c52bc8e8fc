"""Per-kernel averages of the counters in rocprofv3 counter_collection CSVs.   python tools/pmc_read.py <dir-or-csv> [...]"""
import collections, csv, glob, os, sys
for a in sys.argv[1:]:
    fs = [a] if a.endswith(".csv") else glob.glob(os.path.join(a, "**", "*_counter_collection.csv"), recursive=True)
    for f in fs:
        acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("spart::", "")
            acc[k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        print(f)
        for k, v in acc.items():
            if k.startswith("k_"):
                print("  ", k, {c: float(f"{sum(d.values()) / len(d):.4g}") for c, d in v.items()}, "launches", len(next(iter(v.values()))))
