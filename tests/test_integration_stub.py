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
