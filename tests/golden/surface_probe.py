"""The SHAPE of the public surface -- return types, attribute names, array shapes and dtypes of every public callable on the
hot path -- as a JSON-able dict, computed by the SAME code on the reference package (make_golden.py surface -> surface.json)
and on this package (tests/test_surface.py, GPU).  Pure duck typing on the module handed in: nothing of either
implementation is imported here.  A drop-in must agree on all of it (round 6 found `soilwat` returning an array where the
reference returns `SoilOptics`); the few intended differences are listed in tests/test_surface.py.
"""
import io
import warnings
from contextlib import redirect_stdout

import numpy as np


def _arr(x):
    a = np.asarray(x)
    return {"shape": list(a.shape), "dtype": a.dtype.name if a.dtype != object else "object"}


def _obj(o, names):
    out = {"type": type(o).__name__}
    for n in names:
        if not hasattr(o, n):
            out[n] = "MISSING"
            continue
        v = getattr(o, n)
        out[n] = _arr(v) if isinstance(v, (np.ndarray, list, tuple)) or np.ndim(v) else {"scalar": type(v).__name__, "value": (float(v) if isinstance(v, (int, float, np.integer, np.floating)) and not isinstance(v, bool) else str(v))}
    return out


def probe(S, sensors=("Sentinel2A-MSI", "TerraAqua-MODIS")):
    """S: the package (`import SPART`) of either implementation."""
    from importlib import import_module
    bsm, p5d, sailh, smac = (import_module(S.__name__ + "." + m) for m in ("bsm", "prospect_5d", "sailh", "smac"))
    out = {}
    with warnings.catch_warnings(record=True) as wlist, redirect_stdout(io.StringIO()) as so:
        warnings.simplefilter("always")
        out["package_names"] = sorted(n for n in ("SPART", "SpectralBands", "LeafBiology", "SoilParameters", "CanopyStructure", "Angles",
                                                  "AtmosphericProperties", "calculate_ET_radiance", "calculate_spectral_convolution",
                                                  "load_optical_parameters", "load_ET_parameters", "load_sensor_info",
                                                  "set_soil_refl_trans_assumptions", "set_leaf_refl_trans_assumptions") if hasattr(S, n))
        out["submodule_names"] = {m.__name__.split(".")[-1]: sorted(n for n in names if hasattr(m, n)) for m, names in (
            (bsm, ("BSM", "soilwat", "SoilOptics", "SoilParameters", "SoilParametersFromFile")),
            (p5d, ("PROSPECT_5D", "LeafBiology", "LeafOptics", "calculate_tav")),
            (sailh, ("SAILH", "CanopyStructure", "Angles", "CanopyReflectances", "calculate_leafangles")),
            (smac, ("SMAC", "AtmosphericProperties", "AtmosphericOptics", "_calculate_pressure_from_altitude")))}
        op, et = S.load_optical_parameters(), S.load_ET_parameters()
        out["optical_parameters"] = {k: _arr(op[k]) for k in sorted(op) if k in ("nr", "Kab", "Kca", "Kdm", "Kw", "Ks", "Kant", "cbc", "prot", "GSV", "nw")}
        out["ET_parameters"] = {k: _arr(et[k]) for k in sorted(et) if k in ("Ea", "wl_Ea")}
        lb = S.LeafBiology(40, 0.01, 0.02, 0, 10, 10, 1.5)
        out["LeafBiology"] = _obj(lb, ("Cab", "Cdm", "Cw", "Cs", "Cca", "Cant", "N", "PROT", "CBC", "rho_thermal", "tau_thermal"))
        lo = p5d.PROSPECT_5D(lb, op)
        out["PROSPECT_5D"] = _obj(lo, ("refl", "tran", "kChlrel"))
        tv = p5d.calculate_tav(40, np.asarray(op["nr"]))
        out["calculate_tav"] = {"array": _arr(tv), "scalar": type(p5d.calculate_tav(90, 1.4)).__name__ in ("float", "float64")}
        sp_ = S.SoilParameters(0.5, 0, 100, 20, 25, 0.015)
        out["SoilParameters"] = _obj(sp_, ("B", "lat", "lon", "SMp", "SMC", "film", "rdry_set"))
        n0 = len(wlist)
        spd = S.SoilParameters(0.5, 0, 100, 20)
        out["SoilParameters_defaults"] = dict(_obj(spd, ("SMC", "film")), warnings=len(wlist) - n0)
        so_ = bsm.BSM(sp_, op)
        out["BSM"] = _obj(so_, ("refl", "refl_dry"))
        rd = np.asarray(so_.refl_dry).copy()
        sw = bsm.soilwat(rd, op["nw"], op["Kw"], 30.0, 25.0, 0.015)
        out["soilwat"] = dict(_obj(sw, ("refl", "refl_dry")), refl_dry_is_the_input=bool(getattr(sw, "refl_dry", None) is rd))
        n0 = len(wlist)
        sf = bsm.SoilParametersFromFile(rd, 20)
        out["SoilParametersFromFile"] = dict(_obj(sf, ("rdry", "SMp", "SMC", "film", "rdry_set")), warnings=len(wlist) - n0)
        out["BSM_from_file"] = _obj(bsm.BSM(sf, op), ("refl", "refl_dry"))
        cs = S.CanopyStructure(3, -0.35, -0.15, 0.05)
        out["CanopyStructure"] = _obj(cs, ("LAI", "LIDFa", "LIDFb", "q", "nlayers", "nlincl", "nlazi", "lidf"))
        out["calculate_leafangles"] = _arr(sailh.calculate_leafangles(-0.35, -0.15))
        an = S.Angles(40, 0, 0)
        out["Angles"] = _obj(an, ("sol_angle", "obs_angle", "rel_angle"))
        sb = S.SpectralBands()
        out["SpectralBands"] = _obj(sb, ("wlP", "wlE", "WlF", "wlO", "wlT", "wlS", "wlPAR", "nwlP", "nwlT", "IwlP", "IwlT"))
        lo2 = S.set_leaf_refl_trans_assumptions(lo, lb, sb)
        so2 = S.set_soil_refl_trans_assumptions(so_, sb)
        out["set_assumptions"] = {"leaf_same_object": lo2 is lo, "soil_same_object": so2 is so_, "leaf": _obj(lo2, ("refl", "tran", "kChlrel")),
                                  "soil": _obj(so2, ("refl", "refl_dry"))}
        out["SAILH"] = _obj(sailh.SAILH(so2, lo2, cs, an), ("rso", "rdo", "rsd", "rdd"))
        try:
            sailh.SAILH(so2, p5d.PROSPECT_5D(lb, op), cs, an)
            out["SAILH_short_leaf"] = "no error"
        except Exception as e:      # noqa: BLE001
            out["SAILH_short_leaf"] = type(e).__name__
        at = S.AtmosphericProperties(0.325, 0.35, 1.41, Pa=1013.25)
        out["AtmosphericProperties"] = _obj(at, ("aot550", "uo3", "uh2o", "Pa"))
        out["AtmosphericProperties_default_Pa"] = _obj(S.AtmosphericProperties(0.3, 0.3, 1.4), ("Pa",))
        out["AtmosphericProperties_altitude"] = _obj(S.AtmosphericProperties(0.3, 0.3, 1.4, alt_m=1500.0, temp_k=285.0), ("Pa",))
        try:
            S.load_sensor_info("Sentinel9Z")
            out["unknown_sensor"] = "no error"
        except Exception as e:      # noqa: BLE001
            out["unknown_sensor"] = type(e).__name__
        for sensor in sensors:
            si = S.load_sensor_info(sensor)
            d = {"sensorinfo": {k: (_arr(si[k]) if k != "SMAC_coef" else {"n_keys": len(si[k])}) for k in sorted(si)
                                if k in ("wl_smac", "band_id_smac", "SMAC_coef", "wl_srf_smac", "p_srf_smac")}}
            d["SMAC"] = _obj(smac.SMAC(an, at, si["SMAC_coef"]), ("Ta_s", "Ta_o", "Tg", "Ra_dd", "Ra_so", "Ta_ss", "Ta_sd", "Ta_oo", "Ta_do"))
            ra = S.calculate_ET_radiance(et["Ea"], 100, 40)
            d["calculate_ET_radiance"] = _arr(ra)
            d["calculate_spectral_convolution"] = _arr(S.calculate_spectral_convolution(et["wl_Ea"], ra, si))
            sp = S.SPART(S.SoilParameters(0.5, 0, 100, 20, 25, 0.015), S.LeafBiology(40, 0.01, 0.02, 0, 10, 10, 1.5), S.CanopyStructure(3, -0.35, -0.15, 0.05),
                         S.AtmosphericProperties(0.325, 0.35, 1.41, Pa=1013.25), S.Angles(40, 0, 0), sensor, 100)
            d["SPART_attributes_before_run"] = sorted(n for n in ("soilpar", "leafbio", "canopy", "atm", "angles", "sensor", "DOY", "spectral", "optipar", "ETpar", "sensorinfo")
                                                     if hasattr(sp, n))
            df = sp.run()
            d["run"] = {"type": type(df).__name__, "columns": list(df.columns), "dtypes": [str(t) for t in df.dtypes], "index": _arr(df.index.to_numpy()),
                        "index_equals_wl_smac": bool(np.array_equal(df.index.to_numpy(), np.asarray(si["wl_smac"]).T[0])),
                        "Band_equals_band_id": list(df["Band"]) == list(si["band_id_smac"])}
            d["run_attributes"] = _obj(sp, ("R_TOC", "R_TOA", "L_TOA", "_La"))
            d["run_leafopt"] = _obj(sp.leafopt, ("refl", "tran", "kChlrel"))
            d["run_soilopt"] = _obj(sp.soilopt, ("refl", "refl_dry"))
            d["run_canopyopt"] = _obj(sp.canopyopt, ("rso", "rdo", "rsd", "rdd"))
            d["run_atmopt"] = _obj(sp.atmopt, ("Ta_s", "Ta_o", "Tg", "Ra_dd", "Ra_so", "Ta_ss", "Ta_sd", "Ta_oo", "Ta_do"))
            dbg = sp.run(debug=True)
            d["run_debug"] = {"columns": list(dbg.columns), "dtypes": [str(t) for t in dbg.dtypes]}
            out[sensor] = d
        n0 = len(so.getvalue())
        p5d.PROSPECT_5D(S.LeafBiology(40, 0.01, 0.02, 0, 10, 10, 1.5, PROT=0.001, CBC=0.009), op)
        out["PRO_warning_printed"] = len(so.getvalue()) > n0
    return out
