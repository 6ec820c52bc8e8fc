"""Export the reference's static spectral tables to a neutral .npz (run once, in the build container).

Reads the pickles under /root/reference/src/SPART/{model_parameters,sensor_information}
(loaded through the reference's own loaders, SPART.py:399-424) and writes
``spart-python_amd/spart_amd/data/spart_tables.npz`` with ONLY the keys the hot path reads
(SURVEY.md §2 rows 7-8).  The GPU box never unpickles anything.

Layout of the .npz
  wl                      (2001,)  f64   400..2400 nm
  nr,Kab,Kca,Kdm,Kw,Ks,Kant,cbc,prot,nw,Ea   (2001,) f64
  GSV                     (2001,3) f64
  sensors                 (9,)     str   sensor names
  <sensor>/wl_smac        (nb,)    f64   band centres
  <sensor>/wl_smac_is_int ()       bool  pickle dtype was integer
  <sensor>/band_id        (nb,)    str
  <sensor>/coef           (48,nb)  f64   SMAC coefficients, rows ordered as COEF_NAMES
  <sensor>/coef_dtype     (48,)    str   dtype each row had in the pickle (f32 for Sentinel-2)
  <sensor>/wl_srf,p_srf   (nsrf,nb) f64  spectral response function (NaN padded)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _ref_import import import_reference  # noqa: E402

COEF_NAMES = [
    "ah2o", "nh2o", "ao3", "no3", "ao2", "no2", "po2", "aco2", "nco2", "pco2",
    "ach4", "nch4", "pch4", "ano2", "nno2", "pno2", "aco", "nco", "pco",
    "a0s", "a1s", "a2s", "a3s", "a0T", "a1T", "a2T", "a3T", "taur",
    "a0taup", "a1taup", "wo", "gc", "a0P", "a1P", "a2P", "a3P", "a4P",
    "Rest1", "Rest2", "Rest3", "Rest4", "Resr1", "Resr2", "Resr3",
    "Resa1", "Resa2", "Resa3", "Resa4",
]
SENSORS = [
    "TerraAqua-MODIS", "LANDSAT4-TM", "LANDSAT5-TM", "LANDSAT7-ETM", "LANDSAT8-OLI",
    "Sentinel2A-MSI", "Sentinel2B-MSI", "Sentinel3A-OLCI", "Sentinel3B-OLCI",
]


def main():
    SPART = import_reference()
    out = {}
    op = SPART.load_optical_parameters()
    out["wl"] = op["wl"][:, 0].astype(np.float64)
    for k in ["nr", "Kab", "Kca", "Kdm", "Kw", "Ks", "Kant", "cbc", "prot", "nw"]:
        out[k] = np.ascontiguousarray(op[k][:, 0], dtype=np.float64)
    out["GSV"] = np.ascontiguousarray(op["GSV"], dtype=np.float64)
    et = SPART.load_ET_parameters()
    assert np.array_equal(et["wl_Ea"][:, 0].astype(np.float64), out["wl"])
    out["Ea"] = np.ascontiguousarray(et["Ea"][:, 0], dtype=np.float64)
    out["sensors"] = np.array(SENSORS)
    out["coef_names"] = np.array(COEF_NAMES)
    for s in SENSORS:
        si = SPART.load_sensor_info(s)
        wl = si["wl_smac"]
        out[f"{s}/wl_smac"] = wl[:, 0].astype(np.float64)
        out[f"{s}/wl_smac_is_int"] = np.array(np.issubdtype(wl.dtype, np.integer))
        out[f"{s}/band_id"] = np.array([str(b) for b in si["band_id_smac"]])
        c = si["SMAC_coef"]
        out[f"{s}/coef"] = np.stack([c[n][0].astype(np.float64) for n in COEF_NAMES])
        out[f"{s}/coef_dtype"] = np.array([str(c[n].dtype) for n in COEF_NAMES])
        out[f"{s}/wl_srf"] = np.ascontiguousarray(si["wl_srf_smac"], dtype=np.float64)
        out[f"{s}/p_srf"] = np.ascontiguousarray(si["p_srf_smac"], dtype=np.float64)
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                       "spart-python_amd", "spart_amd", "data", "spart_tables.npz")
    np.savez_compressed(dst, **out)
    print("wrote", os.path.normpath(dst), os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
