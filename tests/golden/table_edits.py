"""Edits of the model's TABLE arguments, shared by the fixture generator (make_golden.py gen_tables: applied to the
reference's dicts / objects) and by the GPU tests (applied to this package's).  Pure numpy on dicts of arrays: nothing of
either implementation is imported here.

The reference reads its tables from the dicts it is handed at call time (prospect_5d.py:158-167, bsm.py:45, 54-55) and
from the public attributes of the SPART object on every run() (SPART.py:93-95, 181-184, 192, 202, 216, 228), so each of
these edits changes its answer; the fixtures pin by how much.
"""
import numpy as np


def optical_leaf(op):
    """a new dict: chlorophyll absorption x 1.1, refractive index + 0.02 (moves the three Fresnel tables derived from it,
    prospect_5d.py:200-204), protein absorption x 2"""
    out = dict(op)
    out["Kab"] = np.asarray(op["Kab"], dtype=np.float64) * 1.1
    out["nr"] = np.asarray(op["nr"], dtype=np.float64) + 0.02
    out["prot"] = np.asarray(op["prot"], dtype=np.float64) * 2.0
    return out


def optical_leaf_keys_only(op):
    """the leaf edit above in a dict that holds ONLY the nine keys PROSPECT_5D reads"""
    e = optical_leaf(op)
    return {k: e[k] for k in ("nr", "Kdm", "Kab", "Kca", "Kw", "Ks", "Kant", "cbc", "prot")}


def optical_soil(op):
    """a new dict: flat global soil vectors, water absorption x 0.5, water refractive index + 0.01 (moves the three
    water-film tables derived from it, bsm.py:110-119)"""
    out = dict(op)
    out["GSV"] = np.tile(np.array([0.35, 0.12, 0.06]), (2001, 1))
    out["Kw"] = np.asarray(op["Kw"], dtype=np.float64) * 0.5
    out["nw"] = np.asarray(op["nw"], dtype=np.float64) + 0.01
    return out


def upcast_coefs(sensorinfo):
    """float64 SMAC coefficients (the Sentinel-2 pickles hold float32: without this every comparison carries 1e-7)"""
    sensorinfo["SMAC_coef"] = {k: np.asarray(v).astype(np.float64) for k, v in sensorinfo["SMAC_coef"].items()}


def _v_optipar(sp):
    sp.optipar["Kab"] = np.asarray(sp.optipar["Kab"], dtype=np.float64) * 1.1        # a replaced array
    sp.optipar["GSV"] = np.tile(np.array([0.35, 0.12, 0.06]), (2001, 1))
    upcast_coefs(sp.sensorinfo)


def _v_inplace(sp):
    kw = sp.optipar["Kw"]
    kw *= 1.2                                                                         # the SAME array object, edited in place
    upcast_coefs(sp.sensorinfo)


def _v_etpar(sp):
    sp.ETpar["Ea"] = np.asarray(sp.ETpar["Ea"], dtype=np.float64) * 0.5
    upcast_coefs(sp.sensorinfo)


def _v_sensorinfo(sp):
    si = sp.sensorinfo
    upcast_coefs(si)
    si["wl_smac"] = np.asarray(si["wl_smac"], dtype=np.float64) + 3.5                 # fractional centres: 2-point lerp, new index
    si["p_srf_smac"] = np.asarray(si["p_srf_smac"], dtype=np.float64) ** 2            # a narrower response
    si["SMAC_coef"]["taur"] = si["SMAC_coef"]["taur"] * 1.05
    si["SMAC_coef"]["ah2o"] = si["SMAC_coef"]["ah2o"] * 0.9
    si["band_id_smac"] = ["x" + str(b) for b in si["band_id_smac"]]


def _v_upcast(sp):
    upcast_coefs(sp.sensorinfo)


# name -> edit of an object with .optipar / .ETpar / .sensorinfo (the reference's SPART or this package's)
OBJECT_EDITS = {"optipar": _v_optipar, "inplace": _v_inplace, "etpar": _v_etpar, "sensorinfo": _v_sensorinfo,
                "upcast": _v_upcast}
