"""Drop-in alias: ``import SPART`` resolves to the MI355X evaluator with the reference's names
(``SPART.SPART``, ``SPART.LeafBiology`` ...; submodules ``SPART.bsm``, ``SPART.prospect_5d``,
``SPART.sailh``, ``SPART.smac`` as in src/SPART/)."""
import sys
import types

import spart_amd as _impl
from spart_amd import *  # noqa: F401,F403
from spart_amd.api import _calculate_pressure_from_altitude  # noqa: F401


def _alias(name, names):
    m = types.ModuleType(f"SPART.{name}")
    for n in names:
        setattr(m, n, getattr(_impl.api, n))
    sys.modules[f"SPART.{name}"] = m
    return m


# `from SPART.SPART import SpectralBands` (tests/benchmarks/test_benchmarks.py:7): the module of that name stays
# importable although, as in the reference (src/SPART/__init__.py:1), the ATTRIBUTE SPART.SPART is the class
_alias("SPART", ["SPART", "SpectralBands", "calculate_ET_radiance", "calculate_spectral_convolution", "load_optical_parameters",
                 "load_ET_parameters", "load_sensor_info", "set_soil_refl_trans_assumptions", "set_leaf_refl_trans_assumptions"])


bsm = _alias("bsm", ["BSM", "soilwat", "SoilOptics", "SoilParameters", "SoilParametersFromFile"])
prospect_5d = _alias("prospect_5d", ["PROSPECT_5D", "LeafBiology", "LeafOptics", "calculate_tav"])
sailh = _alias("sailh", ["SAILH", "CanopyStructure", "Angles", "CanopyReflectances", "calculate_leafangles"])
smac = _alias("smac", ["SMAC", "AtmosphericProperties", "AtmosphericOptics", "_calculate_pressure_from_altitude"])
