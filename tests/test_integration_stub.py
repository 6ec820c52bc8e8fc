"""The ctypes stub printed in INTEGRATION.md is executed verbatim (against the in-tree library and the loader
dictionaries a reference maintainer would have) and must reproduce the engine's results."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_integration_md_stub_runs():
    import torch
    import spart_amd
    from spart_amd import _lib, workloads

    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.findall(r"```python\n(.*?)```", text, flags=re.S)[0]
    block = block.replace('ctypes.CDLL("libspart_hip.so")', f'ctypes.CDLL("{_lib.LIB_PATH}")')
    ns = {}
    exec(compile(block, "INTEGRATION.md", "exec"), ns)
    sensor = "Sentinel2A-MSI"
    ctx, keep = ns["make_context"](spart_amd.load_optical_parameters(), spart_amd.load_ET_parameters(),
                                   spart_amd.load_sensor_info(sensor))
    P = torch.as_tensor(workloads.lhs_params(1000, "full", seed=12).T.copy(), device="cuda:0")
    got = ns["run_batch"](ctx, 13, P, "float32")
    ref = spart_amd.get_engine(sensor, 0).run(P, "float32")
    torch.cuda.synchronize()
    for g, k in zip(got, ("R_TOC", "R_TOA", "L_TOA")):
        assert torch.equal(g, ref[k]), k
    ns["lib"].spart_ctx_destroy(ctx)


def test_integration_md_sailh_stub_reads_the_canopy_state():
    """the second block of INTEGRATION.md -- SAILH for the reference's own objects through spart_sailh_batch(lidf_in, nlayers) --
    executed verbatim on top of the first: reproduces the REFERENCE's rows of canopy_state.npz for an assigned lidf and
    another layer count (sailh.py:48, 51)."""
    import sys
    import spart_amd
    from spart_amd import _lib
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import canopy_edits
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    ns = {}
    exec(compile(blocks[0].replace('ctypes.CDLL("libspart_hip.so")', f'ctypes.CDLL("{_lib.LIB_PATH}")'), "INTEGRATION.md", "exec"), ns)
    block = next(b for b in blocks if b.lstrip().startswith("def sailh_batch"))
    exec(compile(block, "INTEGRATION.md#sailh", "exec"), ns)
    ctx, keep = ns["make_context"](spart_amd.load_optical_parameters(), spart_amd.load_ET_parameters(), spart_amd.load_sensor_info("Sentinel2A-MSI"))
    fx = np.load(os.path.join(ROOT, "tests", "golden", "canopy_state.npz"))
    S = spart_amd
    op = S.load_optical_parameters()
    lb = S.LeafBiology(40, 0.01, 0.02, 0, 10, 10, 1.5)
    lo = S.set_leaf_refl_trans_assumptions(S.PROSPECT_5D(lb, op), lb, S.SpectralBands())
    so = S.set_soil_refl_trans_assumptions(S.BSM(S.SoilParameters(0.5, 0, 100, 20, 25, 0.015), op), S.SpectralBands())

    class Canopy:                                   # the attributes the reference's SAILH reads, nothing of this package's classes
        pass
    for e in ("table_nlayers24", "uniform", "nlayers7"):
        for i in (0, 3, 9):
            r = fx["sailh/rows"][i]
            c = Canopy()
            c.LAI, c.LIDFa, c.LIDFb, c.q, c.nlayers = r[0], r[1], r[2], r[3], 60
            c.lidf = fx["sailh/none/lidf"][i].reshape(13, 1).copy()          # what the reference's constructor computed for (a, b)
            canopy_edits.CANOPY_EDITS[e](c, None)
            out = ns["sailh_batch"](ctx, so, lo, c, S.Angles(r[4], r[5], r[6]))
            for j, v in enumerate(out):
                ref = fx[f"sailh/{e}/probes"][i, j]
                got = v[fx["probe_index"], 0]
                assert np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3)) < 1e-9, (e, i, j)
    ns["lib"].spart_ctx_destroy(ctx)
