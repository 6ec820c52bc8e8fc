"""Per-kernel, per-grid-size average durations from a rocprofv3 --kernel-trace CSV.

    python tools/trace_split.py gpurun_out/<dir>/*/*_kernel_trace.csv [min_calls]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
minc = int(sys.argv[2]) if len(sys.argv) > 2 else 3
agg = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void spart::", "").replace("spart::", "")
    g = (int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]))
    agg[(name, g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= minc:
        print(f"{k[0][:44]:44s} grid {k[1][0]:9d}x{k[1][1]:<3d} n={len(v):4d} avg {sum(v) / len(v):8.3f} ms  min {min(v):8.3f}  "
              f"vgpr {0}")
