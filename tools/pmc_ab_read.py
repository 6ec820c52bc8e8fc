"""Print the per-launch SQ counters of k_bands for the two builds profiled by tools/pmc_ab.sh."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "gpurun_out", "pmc_ab")
kern = sys.argv[2] if len(sys.argv) > 2 else "k_bands<float, 0, 1, false>"
for f in sorted(glob.glob(os.path.join(d, "p*.csv"))):
    rows = defaultdict(dict)            # dispatch id -> counter -> value
    for r in csv.DictReader(open(f)):
        if kern.replace(" ", "") in r["Kernel_Name"].replace(" ", ""):
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = rows[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = sorted(rows)
    for side, sel in (("a", ids[0::2]), ("b", ids[1::2])):
        names = sorted(rows[ids[0]])
        print(side, {n: "%.4g" % (sum(rows[i][n] for i in sel) / len(sel)) for n in names})
