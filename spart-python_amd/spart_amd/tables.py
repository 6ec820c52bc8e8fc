"""Static spectral tables (exported once from the reference's pickles by tools/export_tables.py)."""
import os

import numpy as np

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "spart_tables.npz")

SENSORS = ["TerraAqua-MODIS", "LANDSAT4-TM", "LANDSAT5-TM", "LANDSAT7-ETM", "LANDSAT8-OLI",
           "Sentinel2A-MSI", "Sentinel2B-MSI", "Sentinel3A-OLCI", "Sentinel3B-OLCI"]

COEF_NAMES = [
    "ah2o", "nh2o", "ao3", "no3", "ao2", "no2", "po2", "aco2", "nco2", "pco2",
    "ach4", "nch4", "pch4", "ano2", "nno2", "pno2", "aco", "nco", "pco",
    "a0s", "a1s", "a2s", "a3s", "a0T", "a1T", "a2T", "a3T", "taur",
    "a0taup", "a1taup", "wo", "gc", "a0P", "a1P", "a2P", "a3P", "a4P",
    "Rest1", "Rest2", "Rest3", "Rest4", "Resr1", "Resr2", "Resr3",
    "Resa1", "Resa2", "Resa3", "Resa4",
]

_cache = None


def _npz():
    global _cache
    if _cache is None:
        z = np.load(DATA)
        _cache = {k: z[k] for k in z.files}
    return _cache


def load_optical_parameters():
    """Counterpart of SPART.load_optical_parameters (SPART.py:399-406): dict of (2001,1) arrays
    (GSV is (2001,3)), only the keys the model reads."""
    z = _npz()
    out = {k: z[k][:, None].copy() for k in ("nr", "Kab", "Kca", "Kdm", "Kw", "Ks", "Kant", "cbc", "prot", "nw")}
    out["GSV"] = z["GSV"].copy()
    out["wl"] = z["wl"][:, None].astype(np.uint16)
    return out


def load_ET_parameters():
    """Counterpart of SPART.load_ET_parameters (SPART.py:409-416)."""
    z = _npz()
    return {"Ea": z["Ea"][:, None].copy(), "wl_Ea": z["wl"][:, None].astype(np.uint16)}


def load_sensor_info(sensor):
    """Counterpart of SPART.load_sensor_info (SPART.py:419-424); unknown sensors raise
    FileNotFoundError like the reference's open() of a missing pickle."""
    z = _npz()
    if f"{sensor}/wl_smac" not in z:
        raise FileNotFoundError(f"[Errno 2] No such file or directory: 'sensor_information/{sensor}.pkl'")
    wl = z[f"{sensor}/wl_smac"]
    wl = wl.astype(np.uint16) if bool(z[f"{sensor}/wl_smac_is_int"]) else wl.copy()     # (a private copy either way)
    coef = z[f"{sensor}/coef"]
    return {
        "wl_smac": wl[:, None],
        "band_id_smac": [str(b) for b in z[f"{sensor}/band_id"]],
        "SMAC_coef": {n: coef[i][None, :].copy() for i, n in enumerate(COEF_NAMES)},
        "wl_srf_smac": z[f"{sensor}/wl_srf"].copy(),
        "p_srf_smac": z[f"{sensor}/p_srf"].copy(),
    }
