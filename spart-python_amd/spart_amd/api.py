"""The reference's public surface (src/SPART/__init__.py:1-5) on top of the HIP engine.

Same constructor names, positional order and defaults as wirrell/SPART-python; every field
accepts a scalar (reference behaviour) or a length-B array (batched evaluation).  Scalar
calls return objects shaped like the reference's ((n,1) column vectors, a pandas DataFrame
from SPART.run()); batched calls return (B, n) arrays.
"""
import warnings
from dataclasses import dataclass

import numpy as np

from . import engine as _engine
from . import tables as _tables
from .tables import load_ET_parameters, load_optical_parameters, load_sensor_info  # noqa: F401 (re-exported)


_PLAIN = (float, int, np.float64, np.float32, np.int64, np.int32)


def _is_scalar(*vals):
    return all(type(v) in _PLAIN or np.ndim(v) == 0 for v in vals)


def _np(t):
    return t.detach().cpu().numpy()


def _colvec(t, scalar):
    """(B,n) device tensor -> reference layout: (n,1) for a scalar call, (B,n) otherwise."""
    a = _np(t)
    return a[0][:, None].copy() if scalar else a


# ------------------------------------------------------------------------------- leaf
@dataclass
class LeafBiology:
    """prospect_5d.py:19-83 -- positional order Cab, Cdm, Cw, Cs, Cca, Cant, N, then PROT, CBC,
    rho_thermal, tau_thermal."""
    Cab: float
    Cdm: float
    Cw: float
    Cs: float
    Cca: float
    Cant: float
    N: float
    PROT: float = 0.0
    CBC: float = 0.0
    rho_thermal: float = 0.01
    tau_thermal: float = 0.01

    def columns(self):
        return [self.Cab, self.Cdm, self.Cw, self.Cs, self.Cca, self.Cant, self.N, self.PROT, self.CBC]


@dataclass
class LeafOptics:
    """prospect_5d.py:86-114"""
    refl: np.ndarray
    tran: np.ndarray
    kChlrel: np.ndarray


_PRO_WARNING = ("WARNING: When setting PROT and/or CBC > 0. we\n"
                "assume that PROSPECT-PRO was called. Cdm will be\n"
                "therefore set to zero (Cdm = PROT + CBC)")


def _pro_warning(leafbio):
    prot, cbc, cdm = (np.asarray(x, dtype=np.float64) for x in (leafbio.PROT, leafbio.CBC, leafbio.Cdm))
    if np.any(((prot > 0.0) | (cbc > 0.0)) & (cdm > 0)):
        print(_PRO_WARNING)          # prospect_5d.py:148-155 (once per call, not per sample)


def calculate_tav(alpha, nr):
    """prospect_5d.py:249-311, float64 on the host through the library's own routine (spart_calculate_tav: the one the
    context derives its interface tables from); ``nr`` scalar or array, result of the same shape."""
    import ctypes
    from . import _lib
    shape = np.shape(nr)
    a = np.ascontiguousarray(np.asarray(nr, dtype=np.float64).reshape(-1))
    out = np.empty_like(a)
    dp = ctypes.POINTER(ctypes.c_double)
    lib = _lib.load()
    rc = lib.spart_calculate_tav(float(alpha), a.ctypes.data_as(dp), a.size, out.ctypes.data_as(dp))
    if rc != 0:
        raise RuntimeError(lib.spart_last_error(None).decode())
    return float(out[0]) if shape == () else out.reshape(shape)


def PROSPECT_5D(leafbio, optical_params=None, dtype="float64", device=None):
    """prospect_5d.py:117-246.  The nine spectral tables are read from ``optical_params`` as the reference reads them
    (prospect_5d.py:158-167; None = load_optical_parameters()): the device context is chosen by their content."""
    _pro_warning(leafbio)
    eng = _engine.get_engine(None, device, optical_params=optical_params, need=_engine.LEAF_KEYS)
    refl, tran, kchl = eng.prospect(leafbio.columns(), dtype)
    sc = _is_scalar(*leafbio.columns())
    return LeafOptics(_colvec(refl, sc), _colvec(tran, sc), _colvec(kchl, sc))


# ------------------------------------------------------------------------------- soil
class SoilOptics:
    """bsm.py:131-152"""

    def __init__(self, refl, refl_dry):
        self.refl = refl
        self.refl_dry = refl_dry


class SoilParameters:
    """bsm.py:229-287"""

    def __init__(self, B, lat, lon, SMp, SMC=None, film=None):
        self.B = B
        self.lat = lat
        self.lon = lon
        self.SMp = SMp
        if SMC is None:
            warnings.warn("BSM soil model: SMC not supplied, set to default of 25 %")
            self.SMC = 25
        else:
            self.SMC = SMC
        if film is None:
            warnings.warn("BSM soil model: water film optical thickness not supplied, set to default of 0.0150 cm")
            self.film = 0.0150
        else:
            self.film = film
        self.rdry_set = False

    def columns(self):
        return [self.B, self.lat, self.lon, self.SMp, self.SMC, self.film]


class SoilParametersFromFile:
    """bsm.py:155-226: the dry spectrum as an array ((2001,), (2001,1) or (B,2001)) or the path of a JPL / ASTER
    spectral-library text file (host-side parsing; the spectrum then enters the path like any user rdry)."""

    def __init__(self, soil_file, SMp, SMC=None, film=None):
        if SMC is None:
            warnings.warn("BSM soil model: SMC not supplied, set to default of 25 %")
            self.SMC = 25
        else:
            self.SMC = SMC
        if film is None:
            warnings.warn("BSM soil model: water film optical thickness not supplied,")
            warnings.warn("\t set to default of 0.0150 cm")
            self.film = 0.0150
        else:
            self.film = film
        if isinstance(soil_file, np.ndarray):
            self.rdry = soil_file
        else:
            self.rdry = self._load_jpl_soil_refl(soil_file)
        self.SMp = SMp
        self.rdry_set = True

    @staticmethod
    def _load_jpl_soil_refl(file_path):
        """bsm.py:201-226, same result on the same file, NaNs included.  The file: 21 header lines, then
        `wavelength [um] <tab> reflectance [% or fraction]`, wavelengths DESCENDING (the reference slices labels 2401..400
        in file order; an ascending file gives it an empty slice and an all-NaN spectrum, reproduced here with a warning).
        Steps of the reference kept as they are: percent is detected on the whole file (any value > 1); wavelengths
        become nm by a float64 `* 1000`, so a file wavelength counts as "on the 1 nm grid" only if that product is the
        integer exactly; grid wavelengths the file lacks are filled by pandas' `interpolate("linear")`, which is linear
        in ROW POSITION, not in wavelength (np.interp over positions, as pandas does), leaves leading gaps NaN and fills
        trailing gaps with the last value."""
        rows = []
        with open(file_path) as f:
            for i, line in enumerate(f):
                if i < 21 or not line.strip():
                    continue
                a = line.rstrip("\n").split("\t")
                rows.append((float(a[0]), float(a[1]) if len(a) > 1 and a[1].strip() else np.nan))
        if not rows:
            raise ValueError(f"no spectrum rows after the 21 header lines of {file_path}")
        t = np.asarray(rows, dtype=np.float64)
        wl, v = t[:, 0] * 1000, t[:, 1]
        if np.any(v > 1):
            v = v / 100
        dec, inc = bool(np.all(np.diff(wl) <= 0)), bool(np.all(np.diff(wl) >= 0))
        if dec:
            keep = (wl <= 2401) & (wl >= 400)
        elif inc:
            warnings.warn(f"{file_path}: wavelengths ascend; the reference's loader expects descending order and returns NaN")
            keep = np.zeros(len(wl), dtype=bool)
        else:
            raise KeyError(f"{file_path}: wavelengths are not monotonic")
        wl, v = wl[keep], v[keep]
        grid = np.arange(400, 2401, 1).astype(np.float64)
        missing = grid[~np.isin(grid, wl)]
        wl = np.concatenate([wl, missing])
        v = np.concatenate([v, np.full(len(missing), np.nan)])
        o = np.argsort(wl, kind="stable")
        wl, v = wl[o], v[o]
        ok = ~np.isnan(v)
        if ok.any():
            pos = np.arange(len(v), dtype=np.float64)
            filled = np.interp(pos, pos[ok], v[ok])
            filled[: int(np.argmax(ok))] = np.nan
            v = filled
        at = np.searchsorted(wl, grid)
        return v[at][:, None]

    def columns(self):
        return [None, None, None, self.SMp, self.SMC, self.film]


def soilwat(rdry, nw, kw, SMp, SMC, deleff, dtype="float64", device=None):
    """bsm.py:62-128: wet soil reflectance from a dry spectrum with the water tables ``nw`` / ``kw`` GIVEN (None = the
    packaged ones): a device context holding exactly those two tables evaluates it.  Returns what the reference's function
    returns -- ``SoilOptics(rwet, rdry)`` (bsm.py:126-128; its docstring says "np.array", its code returns the object):
    ``refl`` has the shape of ``rdry``, ``refl_dry`` IS the ``rdry`` handed in."""
    op = {}
    if nw is not None:
        op["nw"] = nw
    if kw is not None:
        op["Kw"] = kw
    eng = _engine.get_engine(None, device, optical_params=op or None, need=tuple(op))
    refl, _ = eng.bsm([None, None, None, SMp, SMC, deleff], dtype, rdry=rdry)
    wet = refl.cpu().numpy().reshape(np.shape(rdry)) if np.size(rdry) == 2001 else refl.cpu().numpy()
    return SoilOptics(wet, rdry)


def BSM(soilpar, optical_params=None, dtype="float64", device=None):
    """bsm.py:17-59.  ``optical_params``: GSV, Kw and nw are read from it as the reference reads them (bsm.py:45, 54-55;
    GSV only without a user dry spectrum, bsm.py:42-45); None = load_optical_parameters()."""
    need = ("Kw", "nw") if soilpar.rdry_set else _engine.SOIL_KEYS
    eng = _engine.get_engine(None, device, optical_params=optical_params, need=need)
    rdry = soilpar.rdry if soilpar.rdry_set else None
    refl, dry = eng.bsm(soilpar.columns(), dtype, rdry=rdry)
    sc = _is_scalar(*[c for c in soilpar.columns() if c is not None]) and (rdry is None or np.size(rdry) == 2001)
    return SoilOptics(_colvec(refl, sc), _colvec(dry, sc))


# ------------------------------------------------------------------------------- canopy
class CanopyReflectances:
    """sailh.py:240-272"""

    def __init__(self, rso, rdo, rsd, rdd):
        self.rso = rso
        self.rdo = rdo
        self.rsd = rsd
        self.rdd = rdd


class Angles:
    """sailh.py:275-301"""

    def __init__(self, sol_angle, obs_angle, rel_angle):
        self.sol_angle = sol_angle
        self.obs_angle = obs_angle
        self.rel_angle = rel_angle

    def columns(self):
        return [self.sol_angle, self.obs_angle, self.rel_angle]


def calculate_leafangles(LIDFa, LIDFb, device=None):
    """sailh.py:351-398: (13,1) for scalars, (B,13) for arrays."""
    eng = _engine.get_engine(None, device)
    return _colvec(eng.lidf(LIDFa, LIDFb), _is_scalar(LIDFa, LIDFb))


class CanopyStructure:
    """sailh.py:304-348.  The reference's constructor evaluates ``lidf = calculate_leafangles(LIDFa, LIDFb)`` once and SAILH
    reads ``canopy.lidf`` and ``canopy.nlayers`` from the object at call time (sailh.py:48, 51) -- never LIDFa / LIDFb
    again.  The same holds here: the distribution is bound to the constructor's (LIDFa, LIDFb) (editing ``LIDFa`` afterwards
    changes nothing, as upstream), ``lidf`` may be assigned ((13,), (13, 1) or (B, 13)) or edited in place, and ``nlayers`` is
    honoured (one integer per call).  ``lidf`` itself is evaluated on first access; until then the kernels derive it from
    the bound (LIDFa, LIDFb) with the same routine (bit-identical)."""

    def __init__(self, LAI, LIDFa, LIDFb, q):
        self.LAI = LAI
        self.LIDFa = LIDFa
        self.LIDFb = LIDFb
        self.q = q
        self.nlayers = 60
        self.nlincl = 13
        self.nlazi = 36
        snap = lambda v: np.array(v, dtype=np.float64, copy=True) if isinstance(v, (np.ndarray, list, tuple)) else v  # noqa: E731
        self._lidf_ab = (snap(LIDFa), snap(LIDFb))      # what sailh.py:348 evaluated lidf from
        self._lidf = None

    @property
    def lidf(self):
        if self._lidf is None:
            self._lidf = calculate_leafangles(*self._lidf_ab)
        return self._lidf

    @lidf.setter
    def lidf(self, value):
        self._lidf = value

    def columns(self):
        """[LAI, LIDFa, LIDFb, q] as SAILH sees them: the (LIDFa, LIDFb) the distribution is bound to (None, None once ``lidf``
        exists as an array: the kernels then take :meth:`lidf_state` instead)."""
        if self._lidf is not None:
            return [self.LAI, None, None, self.q]
        return [self.LAI, self._lidf_ab[0], self._lidf_ab[1], self.q]

    def lidf_state(self):
        """``lidf`` once it exists as an array (assigned, or read -- the caller may have edited it in place), else None."""
        return self._lidf


def _canopy_state(canopy):
    """(columns, lidf | None, nlayers | None) of a canopy object: ours, or any object with the reference's attributes."""
    if isinstance(canopy, CanopyStructure):
        cols, lidf = canopy.columns(), canopy.lidf_state()
    else:
        lidf = getattr(canopy, "lidf", None)
        cols = [canopy.LAI, None if lidf is not None else canopy.LIDFa, None if lidf is not None else canopy.LIDFb, canopy.q]
    nl = getattr(canopy, "nlayers", 60)
    return cols, lidf, (None if (isinstance(nl, (int, np.integer)) and not isinstance(nl, bool) and int(nl) == 60) else nl)


def SAILH(soil, leafopt, canopy, angles, dtype="float64", device=None):
    """sailh.py:14-237"""
    refl = np.asarray(leafopt.refl)
    nband = refl.shape[0] if (refl.ndim == 1 or refl.shape[-1] == 1) else refl.shape[-1]
    if nband != 2162:
        raise RuntimeError(
            "Parameter leafopt.refl must be of len 2162"
            " i.e. include thermal specturm. \n This error"
            " usually occurs if you are feeding the prospect_5d"
            " output directly into the SAILH model with adding"
            "\n the neccessary thermal wavelengths."
        )
    eng = _engine.get_engine(None, device)
    ccols, lidf, nl = _canopy_state(canopy)            # canopy.lidf / canopy.nlayers, read here as sailh.py:48, 51 read them
    out = eng.sailh(leafopt.refl, leafopt.tran, soil.refl, ccols, angles.columns(), dtype, canopy_lidf=lidf, nlayers=nl)
    sc = out[0].shape[0] == 1 and _is_scalar(*[c for c in ccols if c is not None], *angles.columns())
    return CanopyReflectances(*[_colvec(o, sc) for o in out])


# ------------------------------------------------------------------------------- atmosphere
class AtmosphericOptics:
    """smac.py:216-272"""

    def __init__(self, Ta_s, Ta_o, Tg, Ra_dd, Ra_so, Ta_ss, Ta_sd, Ta_oo, Ta_do):
        self.Ta_s = Ta_s
        self.Ta_o = Ta_o
        self.Tg = Tg
        self.Ra_dd = Ra_dd
        self.Ra_so = Ra_so
        self.Ta_ss = Ta_ss
        self.Ta_sd = Ta_sd
        self.Ta_oo = Ta_oo
        self.Ta_do = Ta_do


def _calculate_pressure_from_altitude(alt_m, temp_k):
    """smac.py:320-330"""
    g, M, R0, Pa0 = 9.80665, 0.02896968, 8.314462618, 1013.25
    return Pa0 * np.exp(-(g * np.asarray(alt_m, dtype=np.float64) * M / (np.asarray(temp_k, dtype=np.float64) * R0)))


class AtmosphericProperties:
    """smac.py:275-317"""

    def __init__(self, aot550, uo3, uh2o, Pa=None, alt_m=None, temp_k=None):
        self.aot550 = aot550
        self.uo3 = uo3
        self.uh2o = uh2o
        if Pa is None:
            if alt_m is not None and temp_k is not None:
                self.Pa = _calculate_pressure_from_altitude(alt_m, temp_k)
            else:
                self.Pa = 1013.25
        else:
            self.Pa = Pa

    def columns(self):
        return [self.aot550, self.uo3, self.uh2o, self.Pa]


def SMAC(angles, atm, coefs, device=None):
    """smac.py:14-213.  ``coefs`` may be a sensor name or the SMAC_coef dict of load_sensor_info()."""
    if isinstance(coefs, str):
        eng = _engine.get_engine(coefs, device)
    else:
        eng = _engine_for_coefs(coefs, device)
    out = eng.smac(angles.columns(), atm.columns())
    # reference layout is (1, nb) per field for a scalar call
    return AtmosphericOptics(*[_np(out[f]) for f in _engine.SMAC_FIELDS])


def _engine_for_coefs(coefs, device):
    """The engine for a SMAC_coef dict: the reference's calling convention is SMAC(angles, atm, sensorinfo['SMAC_coef']) with a
    freshly loaded dict each time, so the dict's identity says nothing; the engine is found by the CONTENT of the 48 x nb
    coefficient block (engine.get_engine: content-keyed).  SMAC needs no band centres or response functions: placeholders."""
    nb = int(np.size(coefs[_tables.COEF_NAMES[0]]))
    si = {"wl_smac": np.full((nb, 1), 500.0), "band_id_smac": [""] * nb, "SMAC_coef": coefs,
          "wl_srf_smac": np.full((1, nb), 500.0), "p_srf_smac": np.ones((1, nb))}
    return _engine.get_engine(None, device, sensor_info=si)


# ------------------------------------------------------------------------------- orchestration
class SpectralBands:
    """SPART.py:272-315"""

    def __init__(self):
        self.wlP = np.arange(400, 2401, 1)
        self.wlE = np.arange(400, 751, 1)
        self.WlF = np.arange(640, 851, 1)
        self.wlO = np.arange(400, 2401, 1)
        self.wlT = np.concatenate([np.arange(2500, 15001, 100), np.arange(16000, 50001, 1000)])
        self.wlS = np.concatenate([self.wlO, self.wlT])
        self.wlPAR = np.arange(400, 701, 1)
        self.nwlP = len(self.wlP)
        self.nwlT = len(self.wlT)
        self.IwlP = np.arange(0, self.nwlP, 1)
        self.IwlT = np.arange(self.nwlP, self.nwlP + self.nwlT, 1)


_WLS_DEFAULT = np.concatenate([np.arange(400, 2401, 1), np.arange(2500, 15001, 100), np.arange(16000, 50001, 1000)])
_WLS_DEFAULT.setflags(write=False)


def set_soil_refl_trans_assumptions(soilopt, spectral):
    """SPART.py:427-442: pad the soil spectrum with its 2400 nm value (mutates and returns soilopt)."""
    r = np.asarray(soilopt.refl)
    if r.ndim == 2 and r.shape[1] == 1:
        soilopt.refl = np.concatenate([r, np.repeat(r[spectral.nwlP - 1:spectral.nwlP], spectral.nwlT, axis=0)], axis=0)
    else:
        r = np.atleast_2d(r)
        soilopt.refl = np.concatenate([r, np.repeat(r[:, spectral.nwlP - 1:spectral.nwlP], spectral.nwlT, axis=1)], axis=1)
    return soilopt


def set_leaf_refl_trans_assumptions(leafopt, leafbio, spectral):
    """SPART.py:445-470: pad leaf refl/tran with rho_thermal / tau_thermal (mutates and returns leafopt)."""
    def pad(x, v):
        x = np.asarray(x)
        if x.ndim == 2 and x.shape[1] == 1:
            return np.concatenate([x, np.full((spectral.nwlT, 1), float(np.asarray(v).reshape(-1)[0]))], axis=0)
        x = np.atleast_2d(x)
        vv = np.broadcast_to(np.asarray(v, dtype=np.float64).reshape(-1, 1), (x.shape[0], 1))
        return np.concatenate([x, np.repeat(vv, spectral.nwlT, axis=1)], axis=1)
    leafopt.refl = pad(leafopt.refl, leafbio.rho_thermal)
    leafopt.tran = pad(leafopt.tran, leafbio.tau_thermal)
    return leafopt


def calculate_ET_radiance(Ea, DOY, tts):
    """SPART.py:318-355 (host numpy; the batched path applies the same factor inside the kernels)."""
    b = 2 * np.pi * DOY / 365
    corr = 1.00011 + 0.034221 * np.cos(b) + 0.00128 * np.sin(b) + 0.000719 * np.cos(2 * b) + 0.000077 * np.sin(2 * b)
    return Ea * corr * np.cos(tts * np.pi / 180) / np.pi


def calculate_spectral_convolution(wl_hi, radiation_spectra, sensorinfo):
    """SPART.py:358-396 for the 400..2400 nm 1 nm grid: nearest-wavelength lookup (ties to the lower
    wavelength, NaN -> first entry, exactly what the reference's argmin does) and SRF-weighted mean."""
    wl_hi = np.asarray(wl_hi, dtype=np.float64).reshape(-1)
    v = np.asarray(sensorinfo["wl_srf_smac"], dtype=np.float64)
    idx = np.ceil(v - 0.5) - wl_hi[0]
    idx = np.clip(np.where(np.isnan(v), 0, idx), 0, wl_hi.size - 1).astype(np.int64)
    rad = np.asarray(radiation_spectra, dtype=np.float64).reshape(-1)[idx]
    p = sensorinfo["p_srf_smac"]
    return np.sum(rad * p, axis=0) / np.sum(p, axis=0)


class BatchResult(dict):
    """Result of a batched SPART.run(): arrays (B, nb) keyed 'R_TOC', 'R_TOA', 'L_TOA' (+ optional
    materialised fields), with the band table attached."""

    def __init__(self, data, wl, bands):
        super().__init__(data)
        self.wavelengths = wl
        self.bands = bands

    def to_dataframe(self):
        import pandas as pd
        B, nb = self["R_TOC"].shape
        idx = pd.MultiIndex.from_product([range(B), self.wavelengths], names=["sample", "wavelength"])
        cols = {"Band": np.tile(np.asarray(self.bands, dtype=object), B)}
        for k in ("L_TOA", "R_TOA", "R_TOC"):
            cols[k] = np.asarray(self[k]).reshape(-1)
        if "rsoil" in self:
            cols["rsoil"] = np.asarray(self["rsoil"]).reshape(-1)
        return pd.DataFrame(cols, index=idx)


class SPART:
    """SPART.py:35-269.  Stateless per call: every run() evaluates all stages for the current
    parameter objects and tables, which equals what a FRESH reference object returns.  The reference's per-object change
    tracker (stale cached stages after an edit no setter sees, SPART.py:178-209) is deliberately NOT reproduced: see run()."""

    def __init__(self, soilpar, leafbio, canopy, atm, angles, sensor, DOY, dtype="float64", device=None):
        self.soilpar = soilpar
        self.leafbio = leafbio
        self.canopy = canopy
        self.atm = atm
        self.angles = angles
        self.DOY = DOY
        self.dtype = dtype
        self.device = device
        self.spectral = SpectralBands()
        # SPART.py:93-95: public, mutable, and READ BY run() -- an edit of any of the three dicts (or of an array inside
        # one) made before the object's FIRST run() changes it exactly as in the reference; later edits also take effect
        # here, while the reference keeps its cached stages (see run(): stateless by design)
        self.optipar = load_optical_parameters()
        self.ETpar = load_ET_parameters()
        self.sensor = sensor                             # (property: loads sensorinfo; FileNotFoundError for unknown sensors, SPART.py:421-423)

    @property
    def sensor(self):
        return self._sensor

    @sensor.setter
    def sensor(self, sensor):
        # The reference's setter (SPART.py:146-149) only flags the change and keeps the sensorinfo of the constructor, so a
        # re-used object mixes two sensors; a FRESH object -- the parity target -- has the sensorinfo of its sensor.
        self.sensorinfo = load_sensor_info(sensor)
        self._sensor = sensor

    def _columns(self):
        return (self.leafbio.columns() + self.soilpar.columns() + _canopy_state(self.canopy)[0] + self.angles.columns()
                + self.atm.columns() + [self.DOY])

    _LAZY = ("atmopt", "leafopt", "soilopt", "canopyopt")

    def _engine(self):
        """The device context whose tables have the CONTENT of self.optipar / self.ETpar / self.sensorinfo right now
        (SPART.py:181-184, 192, 202, 216, 228).  The three dicts are hashed as they are on every call (engine.raw_digest: ~30 us
        with xxhash); the digest -> (engine, band centres, band ids) resolved by the validating slow path is remembered per
        object, so a loop of run() calls pays for the conversion and validation of the tables once per distinct content."""
        dev = self.device
        if dev is None:                                          # (the memo must follow the CURRENT device, like get_engine)
            import torch
            dev = torch.cuda.current_device() if torch.cuda.is_available() else None
        key = (_engine.raw_digest(self.optipar, self.ETpar, self.sensorinfo), dev)
        memo = self.__dict__.setdefault("_engine_memo", {})
        hit = memo.get(key)
        if hit is None:
            import pandas as pd
            eng = _engine.get_engine(self.sensor, self.device, optical_params=self.optipar, et_params=self.ETpar,
                                     sensor_info=self.sensorinfo)
            wl = np.array(self.sensorinfo["wl_smac"].T[0], copy=True)
            bands = self.sensorinfo["band_id_smac"]
            if len(memo) >= 8:
                memo.pop(next(iter(memo)))
            hit = memo[key] = (eng, wl, bands, pd.Index(wl), np.array(list(bands), dtype=object))
        return hit

    def _run_scalar(self, eng, cols, th, ncol, debug, clidf, nlay):
        """One sample, host scalars in, host columns out: ONE pinned (42,) block up (27 parameters, the two thermal constants, 13
        lidf values), ONE pinned (ncol, nb) block down, both preallocated per object (no per-call tensor allocation), and between
        them one spart_run_batch whose arguments were marshalled once (Engine.prepare)."""
        import torch
        key = (id(eng), self.dtype, len(ncol))
        io = self.__dict__.setdefault("_scalar_io", {})
        b = io.get(key)
        if b is None:
            td = torch.float32 if _engine.DTYPES[self.dtype] == 0 else torch.float64
            hin = torch.empty((42, 1), dtype=torch.float64).pin_memory()
            hout = torch.empty((len(ncol), 1, eng.nb), dtype=td).pin_memory()
            io.clear()                                          # (one staging pair per object: the last configuration's)
            b = io[key] = dict(hin=hin, hin_np=hin.numpy()[:, 0], hout=hout, hout_np=hout.numpy()[:, 0, :],
                               din=torch.empty((42, 1), dtype=torch.float64, device=eng.device),
                               dout=torch.empty((len(ncol), 1, eng.nb), dtype=td, device=eng.device), calls={})
        vals = b["hin_np"]
        for i, c in enumerate(cols):
            vals[i] = 0.0 if c is None else c                   # (None: LIDFa / LIDFb with a given lidf, B / lat / lon with rdry)
        vals[27], vals[28] = th
        if clidf is not None:
            vals[29:42] = np.asarray(clidf, dtype=np.float64).reshape(13)
        din, dout = b["din"], b["dout"]
        sig = (bool(debug), clidf is not None, nlay)
        call = b["calls"].get(sig)
        if call is None:
            fields = ["La", "rsoil"] if debug else ["La"]
            call = b["calls"][sig] = eng.prepare(din[:27], self.dtype, out={k: dout[i] for i, k in enumerate(ncol)},
                                                 rho_thermal=din[27], tau_thermal=din[28], materialize=fields, prune=True,
                                                 canopy_lidf=din[29:42].reshape(1, 13) if clidf is not None else None, nlayers=nlay)
        din.copy_(b["hin"], non_blocking=True)
        call()
        b["hout"].copy_(dout, non_blocking=True)
        torch.cuda.current_stream(eng.device).synchronize()
        host = b["hout_np"].copy()                              # (the staging block is reused by the next call)
        return {k: host[i:i + 1] for i, k in enumerate(ncol)}

    def run(self, debug=False, materialize=False):
        """Returns the reference's DataFrame (columns Band, L_TOA, R_TOA, R_TOC indexed by band
        centre) for scalar parameters, a BatchResult of (B, nb) arrays otherwise.

        ONE spart_run_batch call.  The sensor columns depend on <= 2 nb of the 2162 bands, so a columns-only run
        evaluates just those (spart_materialize.prune_unused_bands: bit-identical columns).  The attributes the reference
        object carries after run() (SPART.py:66-81, 197, 209, 214, 229) -- ``atmopt`` and the full-spectrum
        ``leafopt / soilopt / canopyopt`` -- are computed on FIRST ACCESS from the parameters of this run (one
        spart_smac_batch, resp. one materialising spart_run_batch, cached); ``materialize=True`` computes the spectra
        eagerly in the same call instead.

        STATELESS, deliberately unlike upstream: every run() evaluates all stages from the CURRENT parameter objects and the
        CURRENT optipar / ETpar / sensorinfo, i.e. it returns what a FRESH reference object built from them returns.  The
        reference caches soilopt / leafopt / _La / atmopt behind its _tracker flags (SPART.py:178-209, 226-232), which only the
        property setters flip: after a first run(), ``sp.leafbio.Cab = 60`` or ``sp.optipar["Kab"] *= 1.1`` change NOTHING
        upstream until e.g. ``sp.leafbio = sp.leafbio`` is assigned (tests/golden/canopy_state.npz `stale/` pins that).  Here
        both edits take effect on the next run()."""
        import pandas as pd
        _pro_warning(self.leafbio)
        # the wavelength axis run() interpolates over (SPART.py:220-223 reads self.spectral.wlS): the kernels are built for the
        # reference's grid, so an edited axis is refused, not ignored
        wls = getattr(self.spectral, "wlS", None)
        if not (np.shape(wls) == _WLS_DEFAULT.shape and np.array_equal(wls, _WLS_DEFAULT)):
            raise ValueError("SPART.spectral.wlS differs from the reference's 2162-point grid (SPART.py:303-310): not supported")
        eng, wl, bands, wl_index, bands_arr = self._engine()
        cols = self._columns()
        rdry = self.soilpar.rdry if getattr(self.soilpar, "rdry_set", False) else None
        th = (self.leafbio.rho_thermal, self.leafbio.tau_thermal)
        _, clidf, nlay = _canopy_state(self.canopy)      # canopy.lidf / canopy.nlayers as SAILH reads them (sailh.py:48, 51)
        ncol = ["R_TOC", "R_TOA", "L_TOA", "La"] + (["rsoil"] if debug else [])
        import torch
        fast = (not materialize and rdry is None and _is_scalar(*[c for c in cols if c is not None], *th)
                and not any(torch.is_tensor(c) for c in cols) and (clidf is None or np.size(clidf) == 13))
        if fast:
            out = self._run_scalar(eng, cols, th, ncol, debug, clidf, nlay)
            scalar = True
        else:
            fields = ["La"]
            if debug:
                fields.append("rsoil")
            if materialize:
                fields += _SPECTRA
            # the (B, nb) results share ONE device block, so that they come back in one device-to-host copy
            B = max([int(np.size(c)) if not torch.is_tensor(c) else c.numel() for c in cols if c is not None] + [1])
            if rdry is not None:
                r0 = rdry if torch.is_tensor(rdry) else np.asarray(rdry)
                B = max(B, 1 if (r0.ndim == 1 or (r0.ndim == 2 and r0.shape[1] == 1)) else r0.shape[0])
            if clidf is not None and np.ndim(clidf) == 2 and np.shape(clidf)[1] != 1:
                B = max(B, int(np.shape(clidf)[0]))
            td = torch.float32 if _engine.DTYPES[self.dtype] == 0 else torch.float64
            blk = torch.empty((len(ncol), B, eng.nb), dtype=td, device=eng.device)
            res = eng.run(cols, self.dtype, rho_thermal=th[0], tau_thermal=th[1], materialize=fields, rdry=rdry,
                          prune=not materialize, out={k: blk[i] for i, k in enumerate(ncol)}, canopy_lidf=clidf, nlayers=nlay)
            host = _np(blk)
            out = {k: (host[ncol.index(k)] if k in ncol else _np(v)) for k, v in res.items()}
            scalar = _is_scalar(*[c for c in cols if c is not None]) and out["R_TOC"].shape[0] == 1
        # attributes documented at SPART.py:66-81
        self.R_TOC, self.R_TOA, self.L_TOA = out["R_TOC"], out["R_TOA"], out["L_TOA"]       # (1, nb) for scalars, as upstream (SPART.py:250-252)
        self._La = out["La"][0] if scalar else out["La"]                                   # (nb,) for scalars (SPART.py:183)
        # what the lazy attributes are evaluated from: COPIES of the parameters and the dtype of THIS run (the reference sets the
        # attributes eagerly in run(): mutating an input array in place or changing sp.dtype afterwards must not change them)
        snap = lambda c: None if c is None else (np.array(c, copy=True) if isinstance(c, np.ndarray) else (c.clone() if hasattr(c, "clone") else c))  # noqa: E731
        self.__dict__["_last"] = dict(eng=eng, dtype=self.dtype, cols=[snap(c) for c in cols], th=tuple(snap(t) for t in th),
                                      rdry=snap(rdry), scalar=scalar, clidf=snap(clidf), nlayers=nlay,
                                      angles=[snap(c) for c in self.angles.columns()], atm=[snap(c) for c in self.atm.columns()])
        for k in self._LAZY:
            self.__dict__.pop(k, None)
        if materialize:
            self._set_spectra(out, scalar)
        if not scalar:
            return BatchResult(out, wl, bands)
        data = {"Band": bands_arr.copy(), "L_TOA": out["L_TOA"][0], "R_TOA": out["R_TOA"][0], "R_TOC": out["R_TOC"][0]}   # SPART.py:256-260
        if debug:
            data["rsoil"] = out["rsoil"][0]                                     # SPART.py:262-267
        return pd.DataFrame(data, index=wl_index, copy=False)

    def _set_spectra(self, out, scalar):
        col = (lambda a: a[0][:, None].copy()) if scalar else (lambda a: a)
        self.__dict__["leafopt"] = LeafOptics(col(out["leaf_refl"]), col(out["leaf_tran"]), col(out["leaf_kchl"]))
        self.__dict__["soilopt"] = SoilOptics(col(out["soil_refl"]), col(out["soil_refl_dry"]))
        self.__dict__["canopyopt"] = CanopyReflectances(*[col(out[k]) for k in ("rso", "rdo", "rsd", "rdd")])

    def __getattr__(self, name):
        # only reached when ``name`` is not set: the reference's post-run attributes, computed on first access
        if name in SPART._LAZY:
            last = self.__dict__.get("_last")
            if last is None:
                raise AttributeError(f"{name} is set by run() (SPART.py:197-229)")
            eng = last["eng"]
            if name == "atmopt":
                # atmopt (SPART.py:226-232): the nine SMAC fields, (1, nb) each for scalar inputs (smac.py:209-211), (B, nb) otherwise
                sm = eng.smac(last["angles"], last["atm"])
                self.__dict__["atmopt"] = AtmosphericOptics(*[_np(sm[f]) for f in _engine.SMAC_FIELDS])
            else:
                res = eng.run(last["cols"], last["dtype"], rho_thermal=last["th"][0], tau_thermal=last["th"][1],
                              materialize=_SPECTRA, rdry=last["rdry"], canopy_lidf=last["clidf"], nlayers=last["nlayers"])
                self._set_spectra({k: _np(res[k]) for k in _SPECTRA}, last["scalar"])
            return self.__dict__[name]
        raise AttributeError(f"{type(self).__name__!r} object has no attribute {name!r}")


_SPECTRA = ["leaf_refl", "leaf_tran", "leaf_kchl", "soil_refl", "soil_refl_dry", "rso", "rdo", "rsd", "rdd"]
