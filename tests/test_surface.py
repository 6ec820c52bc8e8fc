"""Drop-in audit of the public surface: tests/golden/surface_probe.py run on THIS package (GPU) against its run on the real
reference (tests/golden/surface.json, make_golden.py surface) -- return types, attribute names, array shapes, dtypes, the
DataFrame's columns / index, exception types, warnings.  Every difference must be in ALLOWED below, with its reason."""
import json
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

# path (dotted) -> reason.  A path ending in ".dtype" allows only the dtype to differ.
ALLOWED = {
    # The Sentinel-2 pickles of the reference store the SMAC coefficients as float32, so under numpy 2 its Ra_dd comes out float32
    # (everything it is combined with is float64, so R_TOA / L_TOA are float64 upstream too); here float64.
    "Sentinel2A-MSI.SMAC.Ra_dd.dtype": "reference: float32 where only float32 coefficients enter (numpy 2 promotion); here float64",
    "Sentinel2A-MSI.run_atmopt.Ra_dd.dtype": "same",
    "*.sensorinfo.SMAC_coef.n_keys": "the reference's pickles carry a 49th row, 'sr', that smac.py:72 has commented out; the packaged dict holds the 48 rows read",
}


def _flatten(d, prefix=""):
    out = {}
    for k, v in d.items():
        p = f"{prefix}.{k}" if prefix else k
        if isinstance(v, dict):
            out.update(_flatten(v, p))
        else:
            out[p] = v
    return out


def _allowed(path):
    import fnmatch
    return any(fnmatch.fnmatch(path, pat) for pat in ALLOWED)


def test_fixture_is_the_reference(tmp_path):
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "surface.json")))
    assert ref["soilwat"]["type"] == "SoilOptics" and ref["soilwat"]["refl_dry_is_the_input"] is True      # (not "np.array" as its docstring says)
    assert ref["SAILH_short_leaf"] == "RuntimeError" and ref["unknown_sensor"] == "FileNotFoundError"
    assert ref["Sentinel2A-MSI"]["run"]["columns"] == ["Band", "L_TOA", "R_TOA", "R_TOC"]
    assert ref["CanopyStructure"]["lidf"]["shape"] == [13, 1] and ref["PRO_warning_printed"] is True


@pytest.mark.gpu
def test_public_surface_matches_the_reference():
    import SPART
    import surface_probe
    ref = _flatten(json.load(open(os.path.join(ROOT, "tests", "golden", "surface.json"))))
    got = _flatten(json.loads(json.dumps(surface_probe.probe(SPART))))
    missing = sorted(k for k in ref if k not in got and not _allowed(k))
    extra = sorted(k for k in got if k not in ref and not _allowed(k))
    diff = sorted(k for k in ref if k in got and ref[k] != got[k] and not _allowed(k))
    assert not missing and not extra, (missing[:20], extra[:20])
    assert not diff, [(k, ref[k], got[k]) for k in diff[:30]]
    # the allowances are exercised, not stale: each pattern matches at least one differing path
    import fnmatch
    differing = [k for k in ref if k in got and ref[k] != got[k]] + [k for k in ref if k not in got] + [k for k in got if k not in ref]
    stale = [pat for pat in ALLOWED if not any(fnmatch.fnmatch(k, pat) for k in differing)]
    assert not stale, f"allowances no longer needed: {stale}; differing: {sorted(differing)}"
