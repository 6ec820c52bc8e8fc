"""Occupancy guard (no GPU needed: hipcc cross-compiles): the register budgets the performance numbers rest on.

The band kernel's speed depends on resident waves per SIMD (512 VGPRs / wave budget): 5 for the float32 kernels
(<= 96 VGPRs, no scratch), 3 for the float64 ones (<= 168).  A change that silently pushes a kernel over the edge
would only show up as a slower benchmark; this test makes it a failure on the build machine."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def _hipcc():
    for c in ("/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    return None


@pytest.fixture(scope="module")
def kernel_meta(tmp_path_factory):
    cc = _hipcc()
    if cc is None:
        pytest.skip("hipcc not available")
    import sys
    sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
    import build                      # every translation unit with ITS flags (build.TU_FLAGS)
    meta, cur = {}, None
    for line in (l for f in build.device_asm(str(tmp_path_factory.mktemp("isa"))) for l in open(f)):
        m = re.match(r"\s+\.name:\s+(\S+)", line)
        if m:
            cur = m.group(1)
            meta[cur] = {}
            continue
        m = re.match(r"\s+\.(vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):\s+(\d+)", line)
        if m and cur:
            meta[cur][m.group(1)] = int(m.group(2))
    return meta


def _find(meta, fragment):
    hits = [v for k, v in meta.items() if fragment in k and "vgpr_count" in v]
    assert hits, fragment
    return hits


def test_float32_band_kernels_keep_five_waves_per_simd(kernel_meta):
    for frag in ("k_bandsIfLi0ELi1E", "k_bandsIfLi1ELi1E", "k_bandsIfLi2ELi1E"):
        for k in _find(kernel_meta, frag):
            assert k["vgpr_count"] <= 96, (frag, k)
            assert k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0, (frag, k)


def test_float64_band_kernels_keep_three_waves_per_simd(kernel_meta):
    for k in _find(kernel_meta, "k_bandsIdLi0ELi1E"):
        assert k["vgpr_count"] <= 168, k
        assert k["private_segment_fixed_size"] <= 256, k       # a few dozen spilled values, not a spilled loop


def test_column_kernel_keeps_four_waves_per_simd(kernel_meta):
    """k_columns (canopy model at the sensor bands + SMAC + TOC->TOA in one wave): <= 128 VGPRs, at most a handful of
    spilled values (2 measured), never a spilled loop"""
    for k in _find(kernel_meta, "k_columnsI"):
        assert k["vgpr_count"] <= 128 and k["private_segment_fixed_size"] <= 32 and k["vgpr_spill_count"] <= 4, k
