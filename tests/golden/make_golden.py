"""Generate the golden vectors from the REAL reference (run in the build container only).

    python tests/golden/make_golden.py

Imports /root/reference/src/SPART (tools/_ref_import.py), runs the reference's own functions on
the parameter sets below and stores inputs + outputs as .npz next to this script.  The
fixtures are data only; nothing of the reference's source is stored.  The GPU box has no
/root/reference: tests there read the committed .npz files.

Files
  prospect.npz   leaf (n,9) + refl/tran/kChlrel (n,2001)   PROSPECT_5D (prospect_5d.py:117)
  bsm.npz        soil (n,6) + refl/refl_dry (n,2001)       BSM (bsm.py:17)
  sailh.npz      canopy (n,4), angles (n,3), lidf (n,13) + rso/rdo/rsd/rdd (n,2162)
                 with the reference's default leaf/soil fixtures (tests/conftest.py:48-59)
  smac.npz       per sensor: angles, atm + the 9 AtmosphericOptics fields (smac.py:14)
  e2e.npz        full SPART(...).run() rows: defaults x 9 sensors, README/MODIS, PRO/S2B,
                 256 rows of the config-4 LHS (S2A), 64 rows config-5 LHS (S2B), 32 LHS rows MODIS/L7/S3A
  jpl.npz        SoilParametersFromFile(<path>).rdry (2001,1) for the synthetic JPL-format text files of jpl/
                 (written by this script: descending percent with irregular steps, fraction units, a file that
                 starts above 400 nm, an ascending file) (bsm.py:201-226)
  grids.npz      the reference's full unit-test grids (6480 PROSPECT + 8100 SAILH cases): 16 probe bands + the all-band mean of
                 every spectrum (python tests/golden/make_golden.py grids; ~5 min on 8 processes)
  s2_f64.npz     Sentinel-2A/B with the reference's OWN arithmetic but its SMAC coefficients up-cast to float64 IN THE HARNESS
                 (the S2 pickles store them as float32, so the reference's outputs carry ~1e-7 of float32 noise that every
                 comparison with smac.npz / e2e.npz / edge.npz has to allow for): the 12 SMAC rows of smac.npz, defaults,
                 PRO, the 256 config-4 rows, the 64 config-5 rows and the 128 edge rows.  Pins the ALGEBRA of smac_band on
                 the headline sensors at 1e-12 / 1e-10 (SPART.py:228, smac.py:44-92, 100-211)
  tables.npz     the TABLE arguments honoured at call time (table_edits.py): PROSPECT_5D(lb, op') / BSM(sp, op') / soilwat with
                 edited optical_params, and SPART objects whose optipar / ETpar / sensorinfo were edited between
                 construction and run() (SPART.py:93-95 read at :181-184, 192, 202, 216, 228; prospect_5d.py:158-167;
                 bsm.py:45, 54-55): defaults + 8 LHS rows per edit, Sentinel2A (+ MODIS for the sensorinfo edit)
  canopy_state.npz  the canopy STATE SAILH reads from the object at call time (canopy_edits.py: canopy.lidf assigned / edited in
                 place / of another (a, b) / un-normalised / 1-D, canopy.nlayers 1, 7, 30, 120, LIDFa edited after construction;
                 sailh.py:48-54, 93-97, 131-135, 216-219, 348): per edit SAILH(...) on the default optics (16 probe bands + all-band
                 mean) and SPART(...).run(debug=True) for defaults + 8 LHS rows on Sentinel2A (float64 coefficients) and
                 TerraAqua-MODIS; and `stale/`: a second run() of ONE object after an edit of optipar / a leafbio field (the
                 reference's change tracker keeps its first answer, SPART.py:178-209) beside a fresh object's
                 (python tests/golden/make_golden.py canopy_state)
  config2.npz    BASELINE config 2's own workload: the first 32 rows of workloads.lhs_params(10_000, "leaf") (7-D LHS, PROT = CBC = 0)
                 through PROSPECT_5D: leaf (32,9) + refl / tran / kChlrel (32,2001)   (prospect_5d.py:117-246)
  surface.json   the SHAPE of the reference's public surface (surface_probe.py: return types, attribute names, array shapes and dtypes of
                 every public callable on the hot path, S2A + MODIS), for the drop-in audit of tests/test_surface.py
  edge.npz       128 rows of tools/edge_sweep.py's widened ranges with edge values (LAI 0 / 1e-4 / 10, dry soil, N = 1,
                 zero pigments, exact hot spot, grazing angles, PRO leaves), Sentinel2A: P + R_TOC / R_TOA / L_TOA
"""
import io
import itertools
import os
import sys
from contextlib import redirect_stdout
from multiprocessing import Pool

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, HERE)
from _ref_import import import_reference  # noqa: E402

SPART = import_reference()
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd", "spart_amd"))
import workloads  # noqa: E402  (plain module import: does not pull in the package / HIP lib)

from SPART.bsm import BSM, SoilParameters  # noqa: E402
from SPART.prospect_5d import PROSPECT_5D, LeafBiology  # noqa: E402
from SPART.sailh import SAILH, Angles, CanopyStructure  # noqa: E402
from SPART.smac import SMAC, AtmosphericProperties  # noqa: E402

SENSORS = ["TerraAqua-MODIS", "LANDSAT4-TM", "LANDSAT5-TM", "LANDSAT7-ETM", "LANDSAT8-OLI",
           "Sentinel2A-MSI", "Sentinel2B-MSI", "Sentinel3A-OLCI", "Sentinel3B-OLCI"]


def sample_rows(n_total, n=10):
    """Row positions pandas' DataFrame.sample(n, random_state=42) draws (tests/conftest.py:28-30)."""
    import pandas as pd
    return list(pd.DataFrame(index=range(n_total)).sample(n, random_state=42).index)


def prospect_grid():
    """build_PROSPECT_tests.py:38-50; the builder prepends each new row, so file row i is combination n-1-i."""
    Cab = np.arange(10, 85, 10); Cca = np.arange(10, 35, 10); Cw = np.arange(0.02, 0.12, 0.04)
    Cdm = np.arange(0.005, 0.025, 0.01); Cs = np.arange(0, 1.5, 0.5); Cant = np.arange(10, 35, 10)
    N = np.arange(1.0, 3.5, 0.5)
    g = list(itertools.product(Cab, Cdm, Cw, Cs, Cca, Cant, N))
    return g[::-1]


def sailh_grid():
    """build_SAILH_tests.py:87-101."""
    LAI = np.arange(1, 8, 3); a = np.arange(-1, 1, 0.4); b = np.arange(-1, 1, 0.4); q = np.arange(0.01, 0.2, 0.05)
    s = np.arange(0, 75, 30); o = np.arange(0, 75, 30); r = np.arange(0, 180, 80)
    g = list(itertools.product(LAI, a, b, q, s, o, r))
    return g[::-1]


def gen_prospect():
    op = SPART.load_optical_parameters()
    g = prospect_grid()
    rows = [g[i] for i in sample_rows(len(g))] + [g[0], g[-1]]
    leaf = [list(map(float, r)) + [0.0, 0.0] for r in rows]
    # extra: defaults, README (Cdm=10 -> NaNs at 400-410nm), PRO cases, zero-absorption-ish, N=1
    leaf += [[40, 0.01, 0.02, 0, 10, 10, 1.5, 0, 0], [40, 10, 0.02, 0.01, 0, 10, 1.5, 0, 0],
             [40, 0.01, 0.02, 0, 10, 10, 1.5, 0.001, 0.009], [30, 0.0, 0.015, 0.1, 8, 2, 1.8, 0.002, 0.004],
             [0, 0, 0, 0, 0, 0, 1.5, 0, 0], [5, 0.001, 0.001, 0, 1, 0, 1.0, 0, 0],
             [80, 0.02, 0.05, 0.5, 20, 10, 3.0, 0, 0], [10, 0.002, 0.005, 0, 2, 0, 1.0, 0, 0]]
    leaf = np.array(leaf, dtype=np.float64)
    out = {k: [] for k in ("refl", "tran", "kChlrel")}
    for r in leaf:
        with redirect_stdout(io.StringIO()):
            lo = PROSPECT_5D(LeafBiology(*r[:7], PROT=r[7], CBC=r[8]), op)
        out["refl"].append(lo.refl[:, 0]); out["tran"].append(lo.tran[:, 0]); out["kChlrel"].append(lo.kChlrel[:, 0])
    np.savez_compressed(os.path.join(HERE, "prospect.npz"), leaf=leaf, **{k: np.array(v) for k, v in out.items()})
    print("prospect", leaf.shape)


def _leaf_row(r):
    with redirect_stdout(io.StringIO()):
        lo = PROSPECT_5D(LeafBiology(*r[:7], PROT=r[7], CBC=r[8]), SPART.load_optical_parameters())
    return lo.refl[:, 0], lo.tran[:, 0], lo.kChlrel[:, 0]


def gen_config2():
    leaf = workloads.lhs_params(10_000, "leaf")[:32, :9]
    with Pool(8) as pool:
        res = pool.map(_leaf_row, list(leaf))
    np.savez_compressed(os.path.join(HERE, "config2.npz"), leaf=leaf, refl=np.array([r[0] for r in res]),
                        tran=np.array([r[1] for r in res]), kChlrel=np.array([r[2] for r in res]))
    print("config2", leaf.shape)


def gen_bsm():
    op = SPART.load_optical_parameters()
    soil = np.array([[0.5, 0, 100, 20, 25, 0.015], [0.5, 0, 100, 15, 25, 0.015], [0.5, 0, 100, 5, 25, 0.015],
                     [0.5, 0, 100, 3, 25, 0.015], [0.9, 30, 120, 55, 25, 0.015], [0.3, -30, 80, 5.0001, 25, 0.015],
                     [0.7, 10, 90, 40, 30, 0.02], [0.6, -12.5, 111, 27.5, 20, 0.01], [0.45, 20, 95, 80, 25, 0.015],
                     [0.8, 0, 100, 10, 55, 0.005]], dtype=np.float64)
    refl, dry = [], []
    for r in soil:
        so = BSM(SoilParameters(*r), op)
        if isinstance(so, np.ndarray):   # (never: BSM returns SoilOptics also for mu<=0)
            raise RuntimeError
        refl.append(so.refl[:, 0]); dry.append(so.refl_dry[:, 0])
    np.savez_compressed(os.path.join(HERE, "bsm.npz"), soil=soil, refl=np.array(refl), refl_dry=np.array(dry))
    print("bsm", soil.shape)


def default_optics():
    op = SPART.load_optical_parameters()
    lb = LeafBiology(40, 0.01, 0.02, 0, 10, 10, 1.5)
    lo = SPART.set_leaf_refl_trans_assumptions(PROSPECT_5D(lb, op), lb, SPART.SpectralBands())
    so = SPART.set_soil_refl_trans_assumptions(BSM(SoilParameters(0.5, 0, 100, 20, 25, 0.015), op), SPART.SpectralBands())
    return lo, so


def gen_sailh():
    lo, so = default_optics()
    g = sailh_grid()
    rows = [g[i] for i in sample_rows(len(g))] + [g[0], g[-1]]
    rows = [list(map(float, r)) for r in rows]
    # extra: defaults, hot spot exactly (dso == 0) at nadir and off-nadir, small q, small LAI, psi folding
    rows += [[3, -0.35, -0.15, 0.05, 40, 0, 0], [3, -0.35, -0.15, 0.05, 0, 0, 0], [4, 0.2, -0.1, 0.05, 30, 30, 0],
             [2, -0.35, -0.15, 0.01, 60, 30, 160], [0.1, 0.1, 0.2, 0.2, 10, 25, 90], [7, -0.5, 0.3, 0.1, 55, 5, 270],
             [5, 0.5, -0.3, 0.03, 20, 20, 365], [1.5, 0.0, 0.0, 0.15, 45, 15, 180]]
    rows = np.array(rows, dtype=np.float64)
    out = {k: [] for k in ("rso", "rdo", "rsd", "rdd", "lidf")}
    for r in rows:
        cs = CanopyStructure(*r[:4])
        rad = SAILH(so, lo, cs, Angles(*r[4:7]))
        for k in ("rso", "rdo", "rsd", "rdd"):
            out[k].append(getattr(rad, k)[:, 0])
        out["lidf"].append(cs.lidf[:, 0])
    np.savez_compressed(os.path.join(HERE, "sailh.npz"), canopy=rows[:, :4], angles=rows[:, 4:7],
                        leaf_refl=lo.refl[:, 0], leaf_tran=lo.tran[:, 0], soil_refl=so.refl[:, 0],
                        **{k: np.array(v) for k, v in out.items()})
    print("sailh", rows.shape)


SMAC_FIELDS = ["Ta_s", "Ta_o", "Tg", "Ra_dd", "Ra_so", "Ta_ss", "Ta_sd", "Ta_oo", "Ta_do"]


def gen_smac():
    rng = np.random.default_rng(7)
    out = {}
    for s in ["Sentinel2A-MSI", "Sentinel2B-MSI", "TerraAqua-MODIS", "LANDSAT7-ETM", "LANDSAT8-OLI", "Sentinel3A-OLCI"]:
        si = SPART.load_sensor_info(s)
        ang = np.column_stack([rng.uniform(0, 60, 12), rng.uniform(0, 30, 12), rng.uniform(0, 180, 12)])
        atm = np.column_stack([rng.uniform(0.05, 0.5, 12), rng.uniform(0.25, 0.45, 12), rng.uniform(0.5, 4, 12),
                               rng.uniform(950, 1030, 12)])
        ang[0] = [40, 0, 0]; atm[0] = [0.325, 0.35, 1.41, 1013.25]
        ang[1] = [40, 0, 0]; atm[1] = [0.3246, 0.3480, 1.4116, 1013.25]
        ang[2] = [30, 10, 180]; ang[3] = [0, 0, 0]
        res = {f: [] for f in SMAC_FIELDS}
        for a, t in zip(ang, atm):
            ao = SMAC(Angles(*a), AtmosphericProperties(t[0], t[1], t[2], Pa=t[3]), si["SMAC_coef"])
            nb = si["wl_smac"].shape[0]
            for f in SMAC_FIELDS:
                res[f].append(np.broadcast_to(np.asarray(getattr(ao, f), dtype=np.float64), (1, nb))[0].copy())
        out[f"{s}/angles"] = ang; out[f"{s}/atm"] = atm
        for f in SMAC_FIELDS:
            out[f"{s}/{f}"] = np.array(res[f])
    np.savez_compressed(os.path.join(HERE, "smac.npz"), **out)
    print("smac done")


def coef_f64(coef):
    """the reference's SMAC_coef dict with every array up-cast to float64 (a new dict: the reference's data are untouched)"""
    return {k: np.asarray(v).astype(np.float64) for k, v in coef.items()}


def run_row(args):
    row, sensor = args[0], args[1]
    upcast = len(args) > 2 and args[2]
    leaf, soil, can, ang, atm, doy = row[0:9], row[9:15], row[15:19], row[19:22], row[22:26], row[26]
    with redirect_stdout(io.StringIO()):
        sp = SPART.SPART(SoilParameters(*soil), LeafBiology(*leaf[:7], PROT=leaf[7], CBC=leaf[8]),
                         CanopyStructure(*can), AtmosphericProperties(atm[0], atm[1], atm[2], Pa=atm[3]),
                         Angles(*ang), sensor, doy if doy != int(doy) else int(doy))
        if upcast:          # on the OBJECT (SPART.py:95, read at SPART.py:228): float64 coefficients, same code
            sp.sensorinfo["SMAC_coef"] = coef_f64(sp.sensorinfo["SMAC_coef"])
        df = sp.run(debug=True)
    probes = [0, 150, 400, 1000, 1600, 2000, 2100]   # 400, 550, 800, 1400, 2000, 2400 nm, first thermal
    extra = np.concatenate([sp.leafopt.refl[probes, 0], sp.leafopt.tran[probes, 0], sp.soilopt.refl[probes, 0],
                            sp.canopyopt.rso[probes, 0], sp.canopyopt.rdo[probes, 0],
                            sp.canopyopt.rsd[probes, 0], sp.canopyopt.rdd[probes, 0]])
    return (df["R_TOC"].to_numpy(), df["R_TOA"].to_numpy(), df["L_TOA"].to_numpy(), df["rsoil"].to_numpy(),
            np.asarray(sp._La, dtype=np.float64), extra)


def gen_e2e():
    out = {}
    groups = []
    d = workloads.default_row()
    for s in SENSORS:
        groups.append((f"defaults/{s}", s, d))
    readme = workloads.default_row(Cab=40, Cdm=10, Cw=0.02, Cs=0.01, Cca=0, Cant=10, N=1.5, SMp=15,
                                   aot550=0.3246, uo3=0.3480, uh2o=1.4116, Pa=1013.25)
    groups.append(("readme/TerraAqua-MODIS", "TerraAqua-MODIS", readme))
    groups.append(("pro/Sentinel2B-MSI", "Sentinel2B-MSI", workloads.default_row(PROT=0.001, CBC=0.009)))
    groups.append(("lhs_full/Sentinel2A-MSI", "Sentinel2A-MSI", workloads.lhs_params(1_000_000, "full")[:256]))
    groups.append(("lhs_pro/Sentinel2B-MSI", "Sentinel2B-MSI", workloads.lhs_params(1_000_000, "pro")[:64]))
    small = workloads.lhs_params(32, "full", seed=11)
    for s in ["TerraAqua-MODIS", "LANDSAT7-ETM", "Sentinel3A-OLCI", "LANDSAT8-OLI"]:
        groups.append((f"lhs_small/{s}", s, small))
    with Pool(8) as pool:
        for name, sensor, P in groups:
            res = pool.map(run_row, [(r, sensor) for r in P], chunksize=4)
            out[f"{name}/P"] = P
            for j, k in enumerate(["R_TOC", "R_TOA", "L_TOA", "rsoil", "La", "probes"]):
                out[f"{name}/{k}"] = np.array([r[j] for r in res])
            print(name, P.shape, flush=True)
    np.savez_compressed(os.path.join(HERE, "e2e.npz"), **out)


def gen_s2f64():
    """Sentinel-2 with float64 SMAC coefficients (see the file list above): same inputs as smac.npz / e2e.npz / edge.npz."""
    import warnings
    import edge_sweep
    warnings.filterwarnings("ignore")
    out = {}
    sm = np.load(os.path.join(HERE, "smac.npz"))
    for s in ("Sentinel2A-MSI", "Sentinel2B-MSI"):
        si = SPART.load_sensor_info(s)
        coef = coef_f64(si["SMAC_coef"])
        assert all(v.dtype == np.float32 for v in si["SMAC_coef"].values())      # (what this fixture is about)
        ang, atm = sm[f"{s}/angles"], sm[f"{s}/atm"]
        nb = si["wl_smac"].shape[0]
        res = {f: [] for f in SMAC_FIELDS}
        for a, t in zip(ang, atm):
            ao = SMAC(Angles(*a), AtmosphericProperties(t[0], t[1], t[2], Pa=t[3]), coef)
            for f in SMAC_FIELDS:
                v = np.asarray(getattr(ao, f))
                assert v.dtype == np.float64
                res[f].append(np.broadcast_to(v, (1, nb))[0].copy())
        out[f"smac/{s}/angles"] = ang; out[f"smac/{s}/atm"] = atm
        for f in SMAC_FIELDS:
            out[f"smac/{s}/{f}"] = np.array(res[f])
    e2e = np.load(os.path.join(HERE, "e2e.npz"))
    edge = np.load(os.path.join(HERE, "edge.npz"))
    groups = [(n, n.split("/")[1], e2e[n + "/P"]) for n in ("defaults/Sentinel2A-MSI", "defaults/Sentinel2B-MSI", "pro/Sentinel2B-MSI",
                                                            "lhs_full/Sentinel2A-MSI", "lhs_pro/Sentinel2B-MSI")]
    groups.append(("edge/" + edge_sweep.SENSOR, edge_sweep.SENSOR, edge["P"]))
    with np.errstate(all="ignore"), Pool(8) as pool:
        for name, sensor, P in groups:
            res = pool.map(run_row, [(r, sensor, True) for r in P], chunksize=4)
            out[f"{name}/P"] = P
            for j, k in enumerate(["R_TOC", "R_TOA", "L_TOA", "rsoil", "La"]):
                out[f"{name}/{k}"] = np.array([r[j] for r in res])
            ref32 = e2e[name + "/R_TOA"] if not name.startswith("edge") else edge["R_TOA"]
            with np.errstate(all="ignore"):
                dev = np.nanmax(np.abs(out[f"{name}/R_TOA"] - ref32) / np.maximum(np.abs(ref32), 1e-6))
            print(name, P.shape, "R_TOA moved by up to %.2e against the float32-coefficient rows" % dev, flush=True)
    np.savez_compressed(os.path.join(HERE, "s2_f64.npz"), **out)


def run_row_edited(args):
    """run_row on an object edited between construction and run() (table_edits.OBJECT_EDITS)"""
    import table_edits
    row, sensor, edit = args
    leaf, soil, can, ang, atm, doy = row[0:9], row[9:15], row[15:19], row[19:22], row[22:26], row[26]
    with redirect_stdout(io.StringIO()):
        sp = SPART.SPART(SoilParameters(*soil), LeafBiology(*leaf[:7], PROT=leaf[7], CBC=leaf[8]),
                         CanopyStructure(*can), AtmosphericProperties(atm[0], atm[1], atm[2], Pa=atm[3]),
                         Angles(*ang), sensor, int(doy))
        table_edits.OBJECT_EDITS[edit](sp)
        df = sp.run(debug=True)
    return (df["R_TOC"].to_numpy(), df["R_TOA"].to_numpy(), df["L_TOA"].to_numpy(), df["rsoil"].to_numpy(),
            np.asarray(sp._La, dtype=np.float64), np.asarray(df.index, dtype=np.float64), np.array(list(df["Band"]), dtype=str))


def gen_tables():
    """The table arguments the reference honours at call time, through the reference's own calls."""
    import warnings
    import table_edits
    from SPART.bsm import soilwat
    warnings.filterwarnings("ignore")
    out = {}
    op = SPART.load_optical_parameters()
    leaf = np.array([[40, 0.01, 0.02, 0, 10, 10, 1.5, 0, 0], [25, 0.004, 0.012, 0.2, 6, 3, 2.2, 0, 0],
                     [60, 0.0, 0.03, 0.0, 15, 1, 1.2, 0.0015, 0.006], [10, 0.02, 0.05, 0.5, 2, 0, 3.0, 0, 0]], dtype=np.float64)
    for tag, opx in (("leaf", table_edits.optical_leaf(op)), ("leaf_keys_only", table_edits.optical_leaf_keys_only(op))):
        res = {k: [] for k in ("refl", "tran", "kChlrel")}
        for r in leaf:
            with redirect_stdout(io.StringIO()):
                lo = PROSPECT_5D(LeafBiology(*r[:7], PROT=r[7], CBC=r[8]), opx)
            for k in res:
                res[k].append(getattr(lo, k)[:, 0])
        for k, v in res.items():
            out[f"prospect/{tag}/{k}"] = np.array(v)
    assert np.array_equal(out["prospect/leaf/refl"], out["prospect/leaf_keys_only/refl"])
    out["prospect/P"] = leaf
    soil = np.array([[0.5, 0, 100, 20, 25, 0.015], [0.9, 30, 120, 55, 25, 0.015], [0.3, -30, 80, 4, 25, 0.015],
                     [0.7, 10, 90, 40, 30, 0.02]], dtype=np.float64)
    ops = table_edits.optical_soil(op)
    refl, dry = [], []
    for r in soil:
        so = BSM(SoilParameters(*r), ops)
        refl.append(so.refl[:, 0]); dry.append(so.refl_dry[:, 0])
    out["bsm/P"] = soil; out["bsm/refl"] = np.array(refl); out["bsm/refl_dry"] = np.array(dry)
    wl = np.arange(400, 2401, dtype=np.float64)
    rdry = (0.08 + 0.25 * (wl - 400) / 2000 + 0.02 * np.sin(wl / 90.0))[:, None]
    sw = soilwat(rdry, ops["nw"], ops["Kw"], 30.0, 25.0, 0.015)
    out["soilwat/rdry"] = rdry[:, 0]; out["soilwat/refl"] = np.asarray(sw.refl)[:, 0]
    # SPART objects edited between construction and run()
    d = workloads.default_row()
    P = np.concatenate([d, workloads.lhs_params(8, "full", seed=23)])
    groups = [(e, "Sentinel2A-MSI") for e in ("optipar", "inplace", "etpar", "sensorinfo", "upcast")] + [("sensorinfo", "TerraAqua-MODIS")]
    with np.errstate(all="ignore"), Pool(8) as pool:
        for edit, sensor in groups:
            res = pool.map(run_row_edited, [(r, sensor, edit) for r in P], chunksize=2)
            name = f"run/{edit}/{sensor}"
            out[name + "/P"] = P
            for j, k in enumerate(["R_TOC", "R_TOA", "L_TOA", "rsoil", "La"]):
                out[f"{name}/{k}"] = np.array([r[j] for r in res])
            out[name + "/index"] = res[0][5]; out[name + "/Band"] = res[0][6]
            base = np.array([r[1] for r in pool.map(run_row_edited, [(r, sensor, "upcast") for r in P[:2]])])
            dev = np.nanmax(np.abs(out[name + "/R_TOA"][:2] - base) / np.maximum(np.abs(base), 1e-6))
            print(name, P.shape, "R_TOA moved by up to %.2e against the unedited object" % dev, flush=True)
    np.savez_compressed(os.path.join(HERE, "tables.npz"), **out)


def gen_rdry():
    """Full chain with user dry-soil spectra (SoilParametersFromFile with an array, bsm.py:155-199, 42-43)."""
    from SPART.bsm import SoilParametersFromFile
    wl = np.arange(400, 2401, dtype=np.float64)
    spectra = np.stack([0.08 + 0.25 * (wl - 400) / 2000 + 0.02 * np.sin(wl / 90.0),
                        0.35 - 0.1 * np.exp(-((wl - 1900) / 60.0) ** 2) + 0.05 * (wl - 400) / 2000,
                        np.full_like(wl, 0.2)])
    rows = [(0, 20.0, "Sentinel2A-MSI"), (1, 45.0, "Sentinel2A-MSI"), (2, 4.0, "Sentinel2A-MSI"), (0, 30.0, "TerraAqua-MODIS")]
    out = {"spectra": spectra}
    d = workloads.default_row()[0]
    for i, (k, smp, sensor) in enumerate(rows):
        with redirect_stdout(io.StringIO()):
            sp = SPART.SPART(SoilParametersFromFile(spectra[k][:, None].copy(), smp, 25, 0.015),
                             LeafBiology(*d[0:7]), CanopyStructure(*d[15:19]),
                             AtmosphericProperties(d[22], d[23], d[24], Pa=d[25]), Angles(*d[19:22]), sensor, 100)
            df = sp.run(debug=True)
        out[f"{i}/spec"] = np.array(k); out[f"{i}/SMp"] = np.array(smp); out[f"{i}/sensor"] = np.array(sensor)
        for c in ("R_TOC", "R_TOA", "L_TOA", "rsoil"):
            out[f"{i}/{c}"] = df[c].to_numpy()
        out[f"{i}/soil_refl"] = sp.soilopt.refl[:, 0]
    np.savez_compressed(os.path.join(HERE, "rdry.npz"), **out)
    print("rdry", len(rows))


def write_jpl_files():
    """Synthetic spectra in the layout of the JPL / ASTER soil files the reference parses (bsm.py:203-206: 21 header lines,
    then `wavelength [um] <tab> reflectance`), data of this repository's own making."""
    d = os.path.join(HERE, "jpl")
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(11)

    def grid(lo_um, hi_um):
        # 2 nm steps to 0.8 um, then 4 nm, then 10 nm, with a few off-grid points (x.5 nm) as real files have
        w = np.concatenate([np.arange(400, 800, 2), np.arange(800, 2500, 4), np.arange(2500, 14000, 10)]).astype(np.float64)
        w[5::37] += 0.5
        w = w[(w >= lo_um * 1000) & (w <= hi_um * 1000)]
        return w / 1000.0

    def refl(w_um):
        x = w_um * 1000
        return 0.12 + 0.22 * (1 - np.exp(-(x - 400) / 600.0)) - 0.07 * np.exp(-((x - 1900) / 70.0) ** 2) - 0.05 * np.exp(-((x - 1400) / 50.0) ** 2)

    def write(name, w, r, unit):
        head = ["Name: synthetic soil %s" % name, "Type: Soil", "Class: Synthetic", "Subclass: none", "Particle Size: Fine",
                "Sample No.: 0", "Owner: spart-python_amd tests", "Wavelength Range: All", "Origin: generated",
                "Collection Date: N/A", "Description: synthetic spectrum in the JPL text layout", "Geologic age: none",
                "Measurement: Directional (10 Degree) Hemispherical Reflectance", "First Column: X", "Second Column: Y",
                "X Units: Wavelength (micrometers)", "Y Units: Reflectance (%s)" % unit, "First X Value: %g" % w[0],
                "Last X Value: %g" % w[-1], "Number of X Values: %d" % len(w), "Additional Information: none"]
        assert len(head) == 21
        with open(os.path.join(d, name + ".txt"), "w") as f:
            f.write("\n".join(head) + "\n")
            for a, b in zip(w, r):
                f.write("%.4f\t%.4f\n" % (a, b))

    w = grid(0.4, 14.0)[::-1]
    write("descending_percent", w, 100 * refl(w) + rng.normal(0, 0.05, len(w)), "percent")
    write("descending_fraction", w, refl(w), "fraction")
    w2 = grid(0.42, 3.0)[::-1]
    write("starts_at_420nm", w2, 100 * refl(w2), "percent")
    write("ascending_percent", w[::-1], 100 * refl(w[::-1]), "percent")
    return d, ["descending_percent", "descending_fraction", "starts_at_420nm", "ascending_percent"]


def gen_jpl():
    """SoilParametersFromFile with a file path (bsm.py:201-226) on the synthetic files of jpl/."""
    import warnings
    from SPART.bsm import SoilParametersFromFile
    d, names = write_jpl_files()
    out = {}
    warnings.filterwarnings("ignore")
    for n in names:
        try:
            out[n] = np.asarray(SoilParametersFromFile(os.path.join(d, n + ".txt"), 20, 25, 0.015).rdry, dtype=np.float64)
            print(n, out[n].shape, "NaN:", int(np.isnan(out[n]).sum()), out[n][[0, 1, 2, 1000, 2000], 0])
        except Exception as e:                                   # the reference's behaviour on this file IS the fixture
            out[n + "/error"] = np.array(type(e).__name__)
            print(n, "raises", type(e).__name__, e)
    # the full chain on the first file (SPART.py:192-199 with rdry_set), defaults of tests/conftest.py
    dflt = workloads.default_row()[0]
    with redirect_stdout(io.StringIO()):
        sp = SPART.SPART(SoilParametersFromFile(os.path.join(d, names[0] + ".txt"), 20, 25, 0.015), LeafBiology(*dflt[0:7]),
                         CanopyStructure(*dflt[15:19]), AtmosphericProperties(dflt[22], dflt[23], dflt[24], Pa=dflt[25]),
                         Angles(*dflt[19:22]), "Sentinel2A-MSI", 100)
        df = sp.run(debug=True)
    for c in ("R_TOC", "R_TOA", "L_TOA", "rsoil"):
        out["run/" + c] = df[c].to_numpy()
    np.savez_compressed(os.path.join(HERE, "jpl.npz"), **out)


def gen_edge():
    import warnings
    import edge_sweep                      # tools/edge_sweep.py: the generator of the widened-range rows
    P = edge_sweep.draw(32768)[:128]
    warnings.filterwarnings("ignore")
    with np.errstate(all="ignore"), Pool(8) as pool:
        res = pool.map(run_row, [(r, edge_sweep.SENSOR) for r in P], chunksize=4)
    out = {"P": P, "sensor": np.array(edge_sweep.SENSOR)}
    for j, k in enumerate(["R_TOC", "R_TOA", "L_TOA"]):
        out[k] = np.array([r[j] for r in res])
    np.savez_compressed(os.path.join(HERE, "edge.npz"), **out)
    print("edge", P.shape, "non-finite reference entries:", int((~np.isfinite(out["R_TOC"])).sum()))


def _canopy_sailh_row(args):
    """SAILH(...) with the default optics on a canopy object edited after construction (canopy_edits.CANOPY_EDITS)"""
    import canopy_edits
    r, edit, payload = args
    lo, so = _GRID_STATE.get("optics") or _GRID_STATE.setdefault("optics", default_optics())
    cs = CanopyStructure(*r[:4])
    if edit:
        canopy_edits.CANOPY_EDITS[edit](cs, payload)
    rad = SAILH(so, lo, cs, Angles(*r[4:7]))
    sp = [np.asarray(getattr(rad, k), dtype=np.float64).reshape(-1) for k in ("rso", "rdo", "rsd", "rdd")]
    assert all(x.size == 2162 for x in sp)
    return np.array([x[GRID_PROBES_CANOPY] for x in sp]), np.array([x.mean() for x in sp]), np.asarray(cs.lidf, dtype=np.float64).reshape(-1)


def _canopy_run_row(args):
    """SPART(...).run(debug=True) on an object whose canopy was edited after construction"""
    import canopy_edits
    import table_edits
    row, sensor, edit, payload = args
    leaf, soil, can, ang, atm, doy = row[0:9], row[9:15], row[15:19], row[19:22], row[22:26], row[26]
    with redirect_stdout(io.StringIO()):
        cs = CanopyStructure(*can)
        sp = SPART.SPART(SoilParameters(*soil), LeafBiology(*leaf[:7], PROT=leaf[7], CBC=leaf[8]), cs,
                         AtmosphericProperties(atm[0], atm[1], atm[2], Pa=atm[3]), Angles(*ang), sensor, int(doy))
        table_edits.upcast_coefs(sp.sensorinfo)
        if edit:
            canopy_edits.CANOPY_EDITS[edit](cs, payload)
        df = sp.run(debug=True)
    return df["R_TOC"].to_numpy(), df["R_TOA"].to_numpy(), df["L_TOA"].to_numpy(), df["rsoil"].to_numpy()


def _stale_rows(sensor):
    """ONE object, run() twice with an edit in between that no setter sees (SPART.py:178-209: the trackers stay False, the
    cached leafopt / soilopt are reused): -> (first run, second run, a FRESH object built after the same edit)."""
    import table_edits
    d = workloads.default_row()[0]

    def make():
        sp = SPART.SPART(SoilParameters(*d[9:15]), LeafBiology(*d[0:7]), CanopyStructure(*d[15:19]),
                         AtmosphericProperties(d[22], d[23], d[24], Pa=d[25]), Angles(*d[19:22]), sensor, 100)
        table_edits.upcast_coefs(sp.sensorinfo)
        return sp
    out = {}
    cols = ("R_TOC", "R_TOA", "L_TOA")
    for tag, edit in (("optipar_kab", lambda sp: sp.optipar.__setitem__("Kab", np.asarray(sp.optipar["Kab"], dtype=np.float64) * 1.1)),
                      ("leafbio_cab", lambda sp: setattr(sp.leafbio, "Cab", 60.0))):
        with redirect_stdout(io.StringIO()):
            sp = make()
            first = sp.run()
            edit(sp)
            second = sp.run()
            fresh = make()
            edit(fresh)
            fr = fresh.run()
        for name, df in (("first", first), ("second", second), ("fresh", fr)):
            for c in cols:
                out[f"stale/{sensor}/{tag}/{name}/{c}"] = df[c].to_numpy()
        print("stale", tag, "second == first:", bool(np.array_equal(second["R_TOC"].to_numpy(), first["R_TOC"].to_numpy())),
              "fresh moved by %.2e" % np.max(np.abs(fr["R_TOC"].to_numpy() / first["R_TOC"].to_numpy() - 1)), flush=True)
    return out


def gen_surface():
    import json
    import surface_probe
    out = surface_probe.probe(SPART)
    with open(os.path.join(HERE, "surface.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("surface", len(json.dumps(out)), "bytes")


def gen_canopy_state():
    import warnings
    import canopy_edits
    from SPART.sailh import calculate_leafangles
    warnings.filterwarnings("ignore")
    payload = np.asarray(calculate_leafangles(*canopy_edits.OTHER_AB), dtype=np.float64).reshape(-1)
    out = {"other_ab_lidf": payload, "probe_index": np.array(GRID_PROBES_CANOPY)}
    d = workloads.default_row()
    P = np.concatenate([d, workloads.lhs_params(8, "full", seed=31)])
    # SAILH rows: canopy + angles of the same nine rows, and the exact hot spot (dso == 0) + a small-q / low-sun row
    rows = np.concatenate([P[:, 15:22], np.array([[3, -0.35, -0.15, 0.05, 30, 30, 0], [2, 0.2, -0.1, 0.01, 60, 30, 160.0]])])
    out["sailh/rows"] = rows
    out["run/P"] = P
    edits = [None] + list(canopy_edits.CANOPY_EDITS)
    with np.errstate(all="ignore"), Pool(8) as pool:
        for e in edits:
            tag = e or "none"
            res = pool.map(_canopy_sailh_row, [(r, e, payload if e == "other_ab" else None) for r in rows], chunksize=2)
            out[f"sailh/{tag}/probes"] = np.array([r[0] for r in res])
            out[f"sailh/{tag}/means"] = np.array([r[1] for r in res])
            out[f"sailh/{tag}/lidf"] = np.array([r[2] for r in res])
            for sensor in ("Sentinel2A-MSI", "TerraAqua-MODIS"):
                rr = pool.map(_canopy_run_row, [(r, sensor, e, payload if e == "other_ab" else None) for r in P], chunksize=2)
                for j, k in enumerate(["R_TOC", "R_TOA", "L_TOA", "rsoil"]):
                    out[f"run/{tag}/{sensor}/{k}"] = np.array([r[j] for r in rr])
            base = out["run/none/Sentinel2A-MSI/R_TOC"]
            print(tag, "R_TOC moved by up to %.2e against the unedited canopy" %
                  np.max(np.abs(out[f"run/{tag}/Sentinel2A-MSI/R_TOC"] / base - 1)), flush=True)
    out.update(_stale_rows("Sentinel2A-MSI"))
    np.savez_compressed(os.path.join(HERE, "canopy_state.npz"), **out)


GRID_PROBES_LEAF = [0, 50, 100, 150, 200, 250, 280, 300, 350, 400, 570, 800, 1050, 1250, 1540, 2000]    # 400 ... 2400 nm
GRID_PROBES_CANOPY = [0, 50, 150, 250, 280, 300, 350, 400, 570, 800, 1050, 1250, 1540, 2000, 2001, 2161]  # + first / last thermal band


def _grid_leaf_row(r):
    with redirect_stdout(io.StringIO()):
        lo = PROSPECT_5D(LeafBiology(*r), _GRID_STATE["op"])
    sp = [lo.refl[:, 0], lo.tran[:, 0], lo.kChlrel[:, 0]]
    return np.array([s[GRID_PROBES_LEAF] for s in sp]), np.array([s.mean() for s in sp])


def _grid_canopy_row(r):
    lo, so = _GRID_STATE["optics"]
    rad = SAILH(so, lo, CanopyStructure(*r[:4]), Angles(*r[4:7]))
    sp = [getattr(rad, k)[:, 0] for k in ("rso", "rdo", "rsd", "rdd")]
    return np.array([s[GRID_PROBES_CANOPY] for s in sp]), np.array([s.mean() for s in sp])


_GRID_STATE = {}


def _grid_init():
    _GRID_STATE["op"] = SPART.load_optical_parameters()
    _GRID_STATE["optics"] = default_optics()


def gen_grids():
    """The reference's own unit-test grids IN FULL, run through the reference itself: all 6480 PROSPECT cases
    (build_PROSPECT_tests.py:38-50) and all 8100 SAILH cases (build_SAILH_tests.py:87-101, default leaf / soil fixtures).
    The reference's parquet files with the expected values are missing from the snapshot (.MISSING_LARGE_BLOBS); whole
    spectra would be 0.9 GB, so each spectrum is stored as 16 probe bands + its mean over ALL bands (the mean pins every
    band in aggregate).  Rows in itertools.product order (the reverse of the parquet row order, which the builder prepends)."""
    import time
    gl = np.array(prospect_grid()[::-1], dtype=np.float64)
    gc = np.array(sailh_grid()[::-1], dtype=np.float64)
    t0 = time.time()
    with np.errstate(all="ignore"), Pool(8, initializer=_grid_init) as pool:
        rc = pool.map(_grid_canopy_row, list(gc), chunksize=32)
        print("sailh grid", len(rc), "%.0f s" % (time.time() - t0), flush=True)
        rl = pool.map(_grid_leaf_row, list(gl), chunksize=16)
        print("prospect grid", len(rl), "%.0f s" % (time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(HERE, "grids.npz"),
                        leaf_grid=gl, leaf_probe_index=np.array(GRID_PROBES_LEAF), leaf_probes=np.array([r[0] for r in rl]),
                        leaf_means=np.array([r[1] for r in rl]),
                        canopy_grid=gc, canopy_probe_index=np.array(GRID_PROBES_CANOPY), canopy_probes=np.array([r[0] for r in rc]),
                        canopy_means=np.array([r[1] for r in rc]))



if __name__ == "__main__":
    which = sys.argv[1:] or ["prospect", "bsm", "sailh", "smac", "e2e", "rdry", "jpl", "edge", "s2f64", "tables"]
    for w in which:
        globals()["gen_" + w]()
