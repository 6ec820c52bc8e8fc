"""spart_amd: MI355X-native batched SPART evaluator (HIP kernels behind libspart_hip.so).

Public names mirror wirrell/SPART-python's ``SPART`` package (src/SPART/__init__.py:1-5).
"""
from .api import (BSM, PROSPECT_5D, SAILH, SMAC, SPART, Angles, AtmosphericOptics, AtmosphericProperties,  # noqa: F401
                  BatchResult, CanopyReflectances, CanopyStructure, LeafBiology, LeafOptics, SoilOptics,
                  SoilParameters, SoilParametersFromFile, SpectralBands, calculate_ET_radiance,
                  calculate_leafangles, calculate_spectral_convolution, calculate_tav, soilwat, load_ET_parameters,
                  load_optical_parameters, load_sensor_info, set_leaf_refl_trans_assumptions,
                  set_soil_refl_trans_assumptions)
from .engine import Engine, get_engine  # noqa: F401
from .tables import SENSORS  # noqa: F401
from . import workloads  # noqa: F401
from .lut import generate_lut, invert_lut, load_lut, lut_to_parquet  # noqa: F401

__version__ = "0.1.0"
