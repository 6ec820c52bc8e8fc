"""A 1M-row Sentinel-2A look-up table written to disk in chunks (copies overlapped with the kernels), then inverted:
the nearest LUT row for a batch of observed spectra.  Needs an MI355X.

    python examples/lut.py [rows] [directory]
"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "spart-python_amd"))
import torch  # noqa: E402
from spart_amd import generate_lut, get_engine, load_lut, workloads  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(tempfile.gettempdir(), "spart_lut")
P = workloads.lhs_params(rows, "full")                         # (rows, 27): the Latin hypercube of SURVEY.md section 8(d)
generate_lut(P, "Sentinel2A-MSI", out, dtype="float32", chunk=250_000)
meta, params, cols = load_lut(out)
print(meta["rows"], "rows,", meta["bands"], cols["R_TOA"].shape)

# inversion: 4096 "observed" TOA spectra = LUT rows + 1 % noise; which row is nearest to each?
eng = get_engine("Sentinel2A-MSI", 0)
rng = np.random.default_rng(1)
pick = rng.integers(0, rows, 4096)
obs = cols["R_TOA"][pick] * (1 + 0.01 * rng.standard_normal((4096, cols["R_TOA"].shape[1]))).astype(np.float32)
idx, cost = eng.lut_nearest(torch.as_tensor(np.ascontiguousarray(cols["R_TOA"]), device="cuda:0"), torch.as_tensor(obs, device="cuda:0"))
idx = idx.cpu().numpy()
print("recovered the generating row for", float((idx == pick).mean()) * 100, "% of the spectra;",
      "median |LAI error| =", float(np.median(np.abs(params[idx, 15] - params[pick, 15]))))

# the same on several GPUs (one process per GPU; nothing changes at one):
#     python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/lut.py
# with, in the script, torch.distributed.init_process_group("nccl") and
#     generate_lut(P, "Sentinel2A-MSI", out, shard=True)        every rank writes its own rows of the same .npy files, no gather
#     idx, cost = spart_amd.invert_lut(out, obs, column="R_TOA", shard=True)   per-rank search + ONE all_gather of the winners
from spart_amd import invert_lut  # noqa: E402
idx2, cost2 = invert_lut(out, obs, column="R_TOA")              # one process: the whole directory on this GPU
assert np.array_equal(idx2, idx)
