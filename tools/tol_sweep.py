import os, sys, json
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
import torch
from spart_amd import workloads
from spart_amd.engine import Engine
for kind, sensor in (("full", "Sentinel2A-MSI"), ("pro", "Sentinel2B-MSI")):
    P = torch.as_tensor(workloads.lhs_params(1_000_000, kind).T.copy(), device="cuda:0")
    for path in sys.argv[1:]:
        e = Engine(sensor, 0, lib_path=path)
        o64 = {k: v.clone() for k, v in e.run(P, "float64").items()}
        o32 = e.run(P, "float32")
        row = {}
        for k in ("R_TOC", "R_TOA"):
            d = (o32[k].double() - o64[k]).abs(); ref = o64[k].abs()
            r3 = d / ref.clamp_min(1e-3)
            row[k] = dict(max_abs="%.2e" % float(d.max()), max_rel3="%.2e" % float(r3.max()), frac_gt_1e4="%.2e" % float((r3 > 1e-4).double().mean()),
                          frac_gt_1e5="%.2e" % float((r3 > 1e-5).double().mean()))
        print(kind, os.path.basename(path), json.dumps(row), flush=True)
