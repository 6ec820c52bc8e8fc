"""Host-side driver of the HIP kernels: one Engine = one libspart_hip context = one
(device, sensor) pair.  torch tensors are used for device memory and streams only; every
compute call goes through the C ABI (include/spart_hip.h)."""
import collections
import ctypes
import threading

import numpy as np

from . import _lib, tables

DTYPES = {"float32": _lib.SPART_F32, "fp32": _lib.SPART_F32, "f32": _lib.SPART_F32,
          "float64": _lib.SPART_F64, "fp64": _lib.SPART_F64, "f64": _lib.SPART_F64}
SMAC_FIELDS = ["Ta_s", "Ta_o", "Tg", "Ra_dd", "Ra_so", "Ta_ss", "Ta_sd", "Ta_oo", "Ta_do"]   # smac.py:209-211
MATERIALIZE_FIELDS = ["leaf_refl", "leaf_tran", "leaf_kchl", "soil_refl", "soil_refl_dry", "rso", "rdo", "rsd",
                      "rdd", "rsoil", "La", "band_mean"]
_MAT_WIDTH = dict(leaf_refl=_lib.NWLS, leaf_tran=_lib.NWLS, leaf_kchl=_lib.NWL, soil_refl=_lib.NWLS,
                  soil_refl_dry=_lib.NWL, rso=_lib.NWLS, rdo=_lib.NWLS, rsd=_lib.NWLS, rdd=_lib.NWLS)


def _require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("spart_amd needs an AMD GPU (HIP device): the evaluator has no CPU path")
    return torch


def _dp(a):
    return a.ctypes.data_as(_lib.c_dp)


# Row pitch (elements) of the (B,2162) / (B,2001) spectrum arrays: rows padded to a multiple of 64 elements start
# on the 128 B line grid, which nearly doubles the HBM store rate of the materialised spectra on MI355X
# (include/spart_hip.h: spart_ctx_set_row_pitch).  Spectra are returned as [:, :width] views of padded storage.
ROW_PITCH = (2176, 2048)


# ---- the static tables a context is built from (spart_tables): the reference reads them from the dicts it is HANDED at call
# time -- PROSPECT_5D(leafbio, optical_params) prospect_5d.py:158-167, BSM(soilpar, optical_params) bsm.py:45, 54-55,
# SPART.run() self.optipar / self.ETpar / self.sensorinfo SPART.py:93-95, 181-184, 192, 202, 228 -- so an engine is keyed on
# their CONTENT, never on a name or on the identity of a dict.
LEAF_KEYS = ("nr", "Kdm", "Kab", "Kca", "Kw", "Ks", "Kant", "cbc", "prot")       # prospect_5d.py:158-167
SOIL_KEYS = ("GSV", "Kw", "nw")                                                     # bsm.py:45, 54-55
OPTICAL_KEYS = ("nr", "Kab", "Kca", "Kdm", "Kw", "Ks", "Kant", "cbc", "prot", "GSV", "nw")
SENSOR_KEYS = ("wl_smac", "SMAC_coef", "wl_srf_smac", "p_srf_smac")               # SPART.py:216, 228, 376-377

try:                                         # (the digest is a cache key, not a security boundary)
    import xxhash as _xx                     # optional dependency: ~20 us per 230 KB of tables

    def _hasher():
        return _xx.xxh3_128()
    HASHER = "xxhash.xxh3_128"
except ImportError:                          # pragma: no cover
    import hashlib as _hl                    # without xxhash: blake2b, ~0.2 ms per call of a scalar SPART.run() (README states both)

    def _hasher():
        return _hl.blake2b(digest_size=16)
    HASHER = "hashlib.blake2b"


def _digest(arrays):
    h = _hasher()
    for a in arrays:
        h.update(np.asarray(a.shape, dtype=np.int64).tobytes())
        h.update(a.data)
    return h.hexdigest()


def _raw_update(h, v):
    """feed one table value into the hasher WITHOUT converting it: dtype + shape + raw bytes (float64 C-contiguous arrays -- the
    packaged tables -- are hashed in place; anything else through one contiguous copy)"""
    a = v if isinstance(v, np.ndarray) else np.asarray(v)
    if a.dtype == object:
        h.update(repr(a.tolist()).encode())
        return
    if not a.flags.c_contiguous:
        a = np.ascontiguousarray(a)
    h.update(a.dtype.str.encode())
    h.update(np.asarray(a.shape, dtype=np.int64).tobytes())
    h.update(a.data if a.ndim else a.tobytes())


def raw_digest(optical_params, et_params, sensor_info):
    """Digest of the three reference-style dicts AS THEY ARE (no float64 conversion, no validation): the keys the context is
    built from, in a fixed order.  Equal raw content -> equal converted tables -> the same engine, so a caller (SPART.run) may
    key a small cache on it and skip optical_block / sensor_block on the hot path; any edit of a dict or of an array in place
    changes it.  A missing key hashes as such (the slow path then raises the reference's KeyError).
    C-contiguous arrays go to the hasher through the buffer protocol as they are (~0.4 us each + 20 us for the 230 KB); their
    dtypes and sizes are hashed once at the end."""
    h = _hasher()
    up = h.update
    meta = []
    add = meta.append
    nd = np.ndarray

    def feed(d, k):
        v = d.get(k) if d is not None else None
        if v is None:
            up(b"\0missing:" + k.encode())
        elif type(v) is nd and v.flags.c_contiguous and v.dtype != object:
            up(v)
            add(v.dtype.num)
            add(v.size)
        else:
            _raw_update(h, v)

    for k in OPTICAL_KEYS:
        feed(optical_params, k)
    feed(et_params, "Ea")
    feed(et_params, "wl_Ea")
    for k in ("wl_smac", "wl_srf_smac", "p_srf_smac", "band_id_smac"):
        feed(sensor_info, k)
    coefs = sensor_info.get("SMAC_coef") if sensor_info is not None else None
    if coefs is None:
        up(b"\0nocoef")
    else:
        for n in tables.COEF_NAMES:
            feed(coefs, n)
    up(np.array(meta, dtype=np.int64))
    return h.digest()


def optical_block(optical_params=None, et_params=None, need=OPTICAL_KEYS):
    """The twelve float64 host tables of spart_tables from the reference's dicts.  ``optical_params`` /
    ``et_params`` = None means the packaged tables (load_optical_parameters / load_ET_parameters).  A key of ``need`` that
    the dict lacks is the reference's own KeyError; keys the calling entry point never reads fall back to the packaged
    table.  Anything that is not a 2001-point spectrum (GSV: (2001, 3)) -- or an ET wavelength axis that is not the
    400..2400 nm grid the SRF convolution indexes -- is refused with a ValueError: never a silently different answer."""
    z = tables._npz()
    out = {}
    for k in OPTICAL_KEYS:
        if optical_params is None or (k not in optical_params and k not in need):
            v = z[k]
        else:
            v = optical_params[k]            # KeyError like the reference's optical_params["..."]
        a = np.ascontiguousarray(np.asarray(v, dtype=np.float64))
        if k == "GSV":
            if a.shape != (_lib.NWL, 3):
                raise ValueError(f"optical_params['GSV'] has shape {a.shape}, expected ({_lib.NWL}, 3) (bsm.py:45-52)")
        else:
            if a.size != _lib.NWL:
                raise ValueError(f"optical_params[{k!r}] has {a.size} entries, expected the {_lib.NWL} bands 400..2400 nm")
            a = a.reshape(-1)
        out[k] = a
    if et_params is None:
        out["Ea"] = np.ascontiguousarray(z["Ea"], dtype=np.float64)
    else:
        ea = np.ascontiguousarray(np.asarray(et_params["Ea"], dtype=np.float64)).reshape(-1)
        if ea.size != _lib.NWL:
            raise ValueError(f"ETpar['Ea'] has {ea.size} entries, expected {_lib.NWL}")
        wl = np.asarray(et_params["wl_Ea"], dtype=np.float64).reshape(-1) if "wl_Ea" in et_params else None
        if wl is None:
            raise KeyError("wl_Ea")          # SPART.py:184
        if wl.size != _lib.NWL or not np.array_equal(wl, np.arange(400, 2401, dtype=np.float64)):
            raise ValueError("ETpar['wl_Ea'] must be the 1 nm grid 400..2400 nm: the SRF convolution of the device context "
                             "(SPART.py:381-387) indexes that grid")
        out["Ea"] = ea
    return out


def sensor_block(sensor_info):
    """The sensor part of spart_tables from a sensorinfo dict (SPART.py:419-424): float64, validated shapes."""
    for k in SENSOR_KEYS:
        if k not in sensor_info:
            raise KeyError(k)
    wl = np.ascontiguousarray(np.asarray(sensor_info["wl_smac"], dtype=np.float64).reshape(-1))
    nb = wl.shape[0]
    coefs = sensor_info["SMAC_coef"]
    rows = []
    for n in tables.COEF_NAMES:
        r = np.asarray(coefs[n], dtype=np.float64).reshape(-1)       # KeyError like smac.py:44-92
        if r.size != nb:
            raise ValueError(f"SMAC_coef[{n!r}] has {r.size} entries for {nb} sensor bands")
        rows.append(r)
    coef = np.ascontiguousarray(np.stack(rows))
    wsrf = np.ascontiguousarray(np.asarray(sensor_info["wl_srf_smac"], dtype=np.float64))
    psrf = np.ascontiguousarray(np.asarray(sensor_info["p_srf_smac"], dtype=np.float64))
    if wsrf.ndim != 2 or wsrf.shape[1] != nb or psrf.shape != wsrf.shape:
        raise ValueError(f"wl_srf_smac {wsrf.shape} / p_srf_smac {psrf.shape} must both be (nsrf, {nb})")
    return dict(wl=wl, coef=coef, wsrf=wsrf, psrf=psrf)



class Engine:
    def __init__(self, sensor=None, device=0, sensor_info=None, lib_path=None, row_pitch=ROW_PITCH, optical_params=None,
                 et_params=None, need=OPTICAL_KEYS):
        """sensor: a packaged sensor name, or None with ``sensor_info`` = a sensorinfo dict (both None: no sensor).
        optical_params / et_params: the reference's table dicts (None = packaged); the context is built from THEIR content.
        row_pitch: (pitch of 2162-wide rows, pitch of 2001-wide rows) or None for dense arrays."""
        torch = _require_gpu()
        self.lib = _lib.load(lib_path)
        self.torch = torch
        self.device = torch.device("cuda", device)
        self.sensor = sensor
        keep = optical_block(optical_params, et_params, need)
        t = _lib.SpartTables()
        for k, v in keep.items():
            setattr(t, k, _dp(v))
        self.nb = 0
        if sensor is not None or sensor_info is not None:
            si = sensor_info if sensor_info is not None else tables.load_sensor_info(sensor)
            self.sensor_info = si
            sb = sensor_block(si)
            keep.update(sb)
            wl, coef, wsrf, psrf = sb["wl"], sb["coef"], sb["wsrf"], sb["psrf"]
            t.nb, t.wl_smac, t.coef = wl.shape[0], _dp(wl), _dp(coef)
            t.nsrf, t.wl_srf, t.p_srf = wsrf.shape[0], _dp(wsrf), _dp(psrf)
            self.nb = int(wl.shape[0])
            self.wl_smac = np.asarray(si["wl_smac"]).reshape(-1)
            self.band_id = list(si["band_id_smac"]) if "band_id_smac" in si else [""] * self.nb
        self._keep = keep
        ctx = _lib.vp()
        rc = self.lib.spart_ctx_create(ctypes.byref(ctx), device, ctypes.byref(t))
        _lib.check(self.lib, None, rc)
        self.ctx = ctx
        self._ws_buf = {}                         # scratch per torch stream (see _workspace)
        self.calls = collections.Counter()        # C-ABI compute calls issued through this engine, by entry point
        self.row_pitch = {_lib.NWLS: _lib.NWLS, _lib.NWL: _lib.NWL}
        if row_pitch is not None:
            pf, po = int(row_pitch[0]), int(row_pitch[1])
            _lib.check(self.lib, self.ctx, self.lib.spart_ctx_set_row_pitch(self.ctx, pf, po))
            self.row_pitch = {_lib.NWLS: pf, _lib.NWL: po}

    def __del__(self):
        try:
            if getattr(self, "ctx", None):
                self.lib.spart_ctx_destroy(self.ctx)
                self.ctx = None
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return ctypes.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _workspace(self, dt, B):
        """The scratch buffer of the CURRENT torch stream: calls issued on different streams (from one or several host
        threads) run concurrently on the GPU, so each stream gets its own.  (The library would also accept one buffer for
        all of them -- it orders a call after the previous user of its workspace -- but that serialises the streams.)"""
        n = int(self.lib.spart_workspace_bytes(self.ctx, dt, B))
        key = self.torch.cuda.current_stream(self.device).cuda_stream
        buf = self._ws_buf.get(key)
        if buf is None or buf.numel() < n:
            buf = self._ws_buf[key] = self.torch.empty(max(n, 256), dtype=self.torch.uint8, device=self.device)
        return ctypes.c_void_p(buf.data_ptr()), ctypes.c_size_t(buf.numel())

    def release_workspace(self):
        """Drop the scratch buffers the engine keeps between calls (one per stream used; each grows to the largest batch
        seen: ~0.9 KB per sample); the next call allocates what it needs."""
        self._ws_buf = {}

    def to_f64(self, x, B=None):
        """scalar / sequence / numpy / tensor -> contiguous float64 device tensor of length B."""
        torch = self.torch
        if not torch.is_tensor(x):
            x = torch.as_tensor(np.asarray(x, dtype=np.float64))
        x = x.to(device=self.device, dtype=torch.float64).reshape(-1)
        if B is not None and x.numel() != B:
            if x.numel() != 1:
                raise ValueError(f"parameter of length {x.numel()} does not broadcast to batch {B}")
            x = x.expand(B)
        return x.contiguous()

    def columns(self, cols):
        """list of per-parameter values -> (list of (B,) tensors, B).  Host values (scalars, sequences, numpy arrays) travel in
        ONE host-to-device copy of an (n, B) block (a scalar SPART.run() used to issue 29 one-element copies: half its
        0.9 ms); device tensors are used where they are."""
        torch = self.torch
        sizes = [int(np.size(c)) if not torch.is_tensor(c) else c.numel() for c in cols]
        B = max(sizes) if sizes else 1
        host = [i for i, c in enumerate(cols) if not torch.is_tensor(c)]
        out = [None] * len(cols)
        if host:
            blk = np.empty((len(host), B), dtype=np.float64)
            for k, i in enumerate(host):
                a = np.asarray(cols[i], dtype=np.float64).reshape(-1)
                if a.size != B and a.size != 1:
                    raise ValueError(f"parameter of length {a.size} does not broadcast to batch {B}")
                blk[k] = a
            dev = torch.as_tensor(blk).to(self.device)
            for k, i in enumerate(host):
                out[i] = dev[k]
        for i, c in enumerate(cols):
            if out[i] is None:
                out[i] = self.to_f64(c, B)
        return out, B

    def _lidf_rows(self, lidf, B):
        """canopy.lidf as the caller set it -> (B, 13) contiguous float64 device tensor, B possibly raised to its row count.
        Accepted: (13,), the reference's (13, 1) column (sailh.py:396), or (B, 13) rows."""
        torch = self.torch
        x = lidf if torch.is_tensor(lidf) else torch.as_tensor(np.asarray(lidf, dtype=np.float64))
        x = x.to(device=self.device, dtype=torch.float64)
        if x.dim() == 1 or (x.dim() == 2 and x.shape[1] == 1 and x.shape[0] == _lib.NLINCL):
            x = x.reshape(1, -1)
        if x.dim() != 2 or x.shape[1] != _lib.NLINCL:
            raise ValueError(f"canopy.lidf of shape {tuple(np.shape(lidf))}: expected ({_lib.NLINCL},), ({_lib.NLINCL}, 1) or (B, {_lib.NLINCL})")
        if x.shape[0] != B:
            if x.shape[0] == 1:
                x = x.expand(B, _lib.NLINCL)
            elif B != 1:
                raise ValueError(f"canopy.lidf has {x.shape[0]} rows for a batch of {B}")
        return x.contiguous()

    @staticmethod
    def _nlayers(n):
        """canopy.nlayers -> the C ABI's int (0 = default).  The reference slices Pso[0:nl] (sailh.py:216): integers only."""
        if n is None:
            return 0
        import numbers
        if isinstance(n, bool) or not isinstance(n, (numbers.Integral, np.integer)):
            if np.ndim(n) != 0:
                raise ValueError("canopy.nlayers must be ONE integer per call (sailh.py:48)")
            raise TypeError(f"canopy.nlayers must be an integer, got {type(n).__name__} (the reference slices Pso[0:nl], sailh.py:216)")
        n = int(n)
        if n < 1 or n > 1000000:
            raise ValueError(f"canopy.nlayers = {n}: expected 1 ... 1000000")
        return n

    def _ptrs(self, tensors):
        arr = (_lib.vp * len(tensors))(*[t.data_ptr() if t is not None else None for t in tensors])
        return arr

    def _tdtype(self, dt):
        return self.torch.float32 if dt == _lib.SPART_F32 else self.torch.float64

    def _alloc_spec(self, B, width, td):
        """(B, width) spectrum array on this context's row pitch (a view of padded storage when pitch > width)."""
        pitch = self.row_pitch[width]
        return self.torch.empty((B, pitch), dtype=td, device=self.device)[:, :width]

    def _spec(self, x, B, width, dt):
        """input spectra -> (B, width) device tensor of dtype dt on this context's row pitch."""
        torch = self.torch
        if not torch.is_tensor(x):
            x = torch.as_tensor(np.asarray(x))
        x = x.to(device=self.device, dtype=self._tdtype(dt))
        if x.dim() == 2 and x.shape[1] == 1 and x.shape[0] == width:     # reference style (n,1) column
            x = x.reshape(1, width)
        x = x.reshape(-1, x.shape[-1])
        if x.shape[1] != width:
            raise ValueError(f"spectrum of length {x.shape[1]}, expected {width}")
        if x.shape[0] != B:
            if x.shape[0] != 1:
                raise ValueError("spectra do not broadcast to the batch")
            x = x.expand(B, width)
        if x.stride() == (self.row_pitch[width], 1):
            return x
        buf = self._alloc_spec(B, width, x.dtype)
        buf.copy_(x)
        return buf

    # ------------------------------------------------------------------ operators
    def prospect(self, leaf9, dtype="float64", outputs=("refl", "tran", "kChlrel")):
        """PROSPECT_5D for a batch (prospect_5d.py:117-246): leaf9 = [Cab,Cdm,Cw,Cs,Cca,Cant,N,PROT,CBC].
        -> [refl, tran, kChlrel], each (B, 2001); entries not named in ``outputs`` are None (not computed / stored)."""
        dt = DTYPES[dtype]
        cols, B = self.columns(leaf9)
        td = self._tdtype(dt)
        out = [self._alloc_spec(B, _lib.NWL, td) if n in outputs else None for n in ("refl", "tran", "kChlrel")]
        ws, wsn = self._workspace(dt, B)
        self.calls["spart_prospect_batch"] += 1
        rc = self.lib.spart_prospect_batch(self.ctx, dt, B, self._ptrs(cols), *[o.data_ptr() if o is not None else None for o in out],
                                           ws, wsn, self._stream())
        _lib.check(self.lib, self.ctx, rc)
        return out

    def bsm(self, soil6, dtype="float64", rdry=None):
        """BSM (bsm.py:17-128): soil6 = [B,lat,lon,SMp,SMC,film]; rdry = optional (B,2001) dry spectra."""
        dt = DTYPES[dtype]
        if rdry is not None:
            soil6 = [0.0 if c is None else c for c in soil6]
        cols, B = self.columns(soil6)
        rd = None
        if rdry is not None:
            rd0 = rdry if self.torch.is_tensor(rdry) else np.asarray(rdry)
            nrow = 1 if rd0.ndim == 1 or (rd0.ndim == 2 and rd0.shape[1] == 1) else rd0.shape[0]
            B = max(B, nrow)
            cols = [c.expand(B).contiguous() if c.numel() == 1 else c for c in cols]
            rd = self._spec(rdry, B, _lib.NWL, dt)
        td = self._tdtype(dt)
        out = [self._alloc_spec(B, _lib.NWL, td) for _ in range(2)]
        ws, wsn = self._workspace(dt, B)
        self.calls["spart_bsm_batch"] += 1
        rc = self.lib.spart_bsm_batch(self.ctx, dt, B, self._ptrs(cols), rd.data_ptr() if rd is not None else None,
                                      out[0].data_ptr(), out[1].data_ptr(), ws, wsn, self._stream())
        _lib.check(self.lib, self.ctx, rc)
        return out

    def lidf(self, LIDFa, LIDFb):
        cols, B = self.columns([LIDFa, LIDFb])
        out = self.torch.empty((B, _lib.NLINCL), dtype=self.torch.float64, device=self.device)
        self.calls["spart_lidf_batch"] += 1
        rc = self.lib.spart_lidf_batch(self.ctx, B, cols[0].data_ptr(), cols[1].data_ptr(), out.data_ptr(),
                                       self._stream())
        _lib.check(self.lib, self.ctx, rc)
        return out

    def sailh(self, rho, tau, rs, canopy4, angles3, dtype="float64", canopy_lidf=None, nlayers=None):
        """SAILH (sailh.py:14-237) on (B,2162) spectra.  canopy_lidf / nlayers: canopy.lidf ((13,), (13,1) or (B,13)) and
        canopy.nlayers as the reference reads them from the object at call time (sailh.py:48, 51); None = derived from
        LIDFa / LIDFb, 60."""
        dt = DTYPES[dtype]
        nl = self._nlayers(nlayers)
        canopy4 = list(canopy4)
        if canopy_lidf is not None:
            canopy4[1] = 0.0 if canopy4[1] is None else canopy4[1]
            canopy4[2] = 0.0 if canopy4[2] is None else canopy4[2]
        cols, B = self.columns(canopy4 + list(angles3))
        for x in (rho, tau, rs):
            n = x.shape[0] if (hasattr(x, "ndim") and x.ndim == 2 and x.shape[1] != 1) else 1
            B = max(B, n)
        li = None
        if canopy_lidf is not None:
            nrow = 1 if (np.ndim(canopy_lidf) == 1 or np.shape(canopy_lidf)[-1] == 1) else np.shape(canopy_lidf)[0]
            B = max(B, nrow)
            li = self._lidf_rows(canopy_lidf, B)
        cols = [c.expand(B).contiguous() if c.numel() == 1 else c for c in cols]
        rho, tau, rs = (self._spec(x, B, _lib.NWLS, dt) for x in (rho, tau, rs))
        td = self._tdtype(dt)
        out = [self._alloc_spec(B, _lib.NWLS, td) for _ in range(4)]
        ws, wsn = self._workspace(dt, B)
        self.calls["spart_sailh_batch"] += 1
        rc = self.lib.spart_sailh_batch(self.ctx, dt, B, rho.data_ptr(), tau.data_ptr(), rs.data_ptr(),
                                        self._ptrs(cols[:4]), self._ptrs(cols[4:]), li.data_ptr() if li is not None else None,
                                        nl, self._ptrs(out), ws, wsn, self._stream())
        _lib.check(self.lib, self.ctx, rc)
        return out

    def smac(self, angles3, atm4):
        """SMAC (smac.py:14-213): nine (B,nb) float64 tensors in AtmosphericOptics order."""
        cols, B = self.columns(list(angles3) + list(atm4))
        out = [self.torch.empty((B, self.nb), dtype=self.torch.float64, device=self.device) for _ in range(9)]
        ws, wsn = self._workspace(_lib.SPART_F64, B)
        self.calls["spart_smac_batch"] += 1
        rc = self.lib.spart_smac_batch(self.ctx, B, self._ptrs(cols[:3]), self._ptrs(cols[3:]), self._ptrs(out), ws, wsn,
                                       self._stream())
        _lib.check(self.lib, self.ctx, rc)
        return dict(zip(SMAC_FIELDS, out))

    def run(self, params, dtype="float32", rho_thermal=None, tau_thermal=None, materialize=(), out=None,
            prune=False, rdry=None, f32_columns=False, f32_bands=False, lidf="literal", _workspace=None,
            canopy_lidf=None, nlayers=None, _defer=False):
        """SPART(...).run() for every column of ``params`` (SPART.py:162-269).

        params : (27, B) float64 device tensor (rows = spart_amd.workloads.PARAM_NAMES) or a list of 27
                 scalars / arrays.
        materialize : iterable of names from MATERIALIZE_FIELDS to also return (full spectra etc.)
        out : optional dict with preallocated 'R_TOC','R_TOA','L_TOA' (B,nb) tensors and / or preallocated tensors for
              names in ``materialize`` (spectrum arrays on this engine's row pitch, e.g. a previous call's results)
        prune : False (default) evaluates all 2162 bands of every sample; True lets the kernel skip bands
                that no requested output needs (identical columns, much less work)
        rdry : optional (B, 2001) / (2001,) user dry-soil spectra replacing the GSV mixing (bsm.py:42-43);
               the B / lat / lon entries of ``params`` are then ignored (may be None in a list)
        f32_bands : float64 only.  True: float64 columns identical to the float64 mode's over a float32 evaluation of
               the 2162 bands (spart_materialize.f32_bands): reference precision at the float32 mode's speed
        lidf : "literal" (default) = the reference's LIDF fixed-point iteration with its stopping rule and 10-point hot-spot
               panels; "newton" = spart_materialize.fast_prelude: the exact root + 8-point panels, columns within 1e-7
               relative of the default, the prelude kernel twice as fast
        f32_columns : float32 only.  False (default): the bands the sensor columns depend on are re-evaluated in
               float64, the columns are the float64 mode's values rounded to float32; True: columns straight from
               the float32 band arithmetic (spart_materialize.f32_columns)
        canopy_lidf : optional canopy.lidf as the caller set it, (13,), (13,1) or (B,13) (spart_materialize.lidf_in; the
               reference's SAILH reads it from the object, sailh.py:51); the LIDFa / LIDFb entries of ``params`` are then
               ignored (may be None in a list)
        nlayers : optional canopy.nlayers, one integer per call (spart_materialize.nlayers; sailh.py:48); None = 60
        """
        torch = self.torch
        dt = DTYPES[dtype]
        td = self._tdtype(dt)
        col_ptrs = None
        if torch.is_tensor(params) and params.dim() == 2:
            if params.shape[0] != _lib.NPARAM:
                raise ValueError("params must be (27, B)")
            P = params.to(device=self.device, dtype=torch.float64).contiguous()
            B = P.shape[1]
            cols = None
            base = P.data_ptr()                     # row i of the contiguous block: no 27 tensor views, no 27 data_ptr() calls
            col_ptrs = (_lib.vp * _lib.NPARAM)(*[base + 8 * B * i for i in range(_lib.NPARAM)])
        else:
            plist = [0.0 if (p is None and (rdry is not None or (canopy_lidf is not None and i in (16, 17)))) else p
                     for i, p in enumerate(params)]
            if len(plist) != _lib.NPARAM:
                raise ValueError(f"params must have {_lib.NPARAM} entries, got {len(plist)}")
            for i, p in enumerate(plist):
                if p is None:
                    raise ValueError(f"params[{i}] is None (only B / lat / lon with rdry= and LIDFa / LIDFb with canopy_lidf= may be)")
            cols, B = self.columns(plist)
            if rdry is not None:
                r0 = rdry if torch.is_tensor(rdry) else np.asarray(rdry)
                nrow = 1 if (r0.ndim == 1 or (r0.ndim == 2 and r0.shape[1] == 1)) else r0.shape[0]
                if nrow > B:
                    B = nrow
                    cols = [c.expand(B).contiguous() if c.numel() == 1 else c for c in cols]
            if canopy_lidf is not None:
                nrow = 1 if (np.ndim(canopy_lidf) == 1 or np.shape(canopy_lidf)[-1] == 1) else np.shape(canopy_lidf)[0]
                if nrow > B:
                    B = nrow
                    cols = [c.expand(B).contiguous() if c.numel() == 1 else c for c in cols]
        nl = self._nlayers(nlayers)
        li = self._lidf_rows(canopy_lidf, B) if canopy_lidf is not None else None
        th = [None if x is None else self.to_f64(x, B) for x in (rho_thermal, tau_thermal)]
        res = dict(out) if out is not None else {}       # (the caller's dict is not modified)
        for k in ("R_TOC", "R_TOA", "L_TOA"):
            if k not in res:
                res[k] = torch.empty((B, self.nb), dtype=td, device=self.device)
            else:
                self._check_out(k, res[k], (B, self.nb), td)
        mat = None
        rd = None
        if lidf not in ("literal", "newton"):
            raise ValueError("lidf must be 'literal' or 'newton'")
        if materialize or prune or rdry is not None or f32_columns or f32_bands or lidf == "newton" or li is not None or nl:
            mat = _lib.SpartMaterialize()
            mat.lidf_in = li.data_ptr() if li is not None else None
            mat.nlayers = nl
            mat.fast_prelude = 1 if lidf == "newton" else 0
            mat.prune_unused_bands = 1 if prune else 0
            mat.f32_columns = 1 if f32_columns else 0
            mat.f32_bands = 1 if f32_bands else 0
            if rdry is not None:
                rd = self._spec(rdry, B, _lib.NWL, dt)
                mat.rdry_in = rd.data_ptr()
            for name in materialize:
                if name not in MATERIALIZE_FIELDS:
                    raise ValueError(f"unknown materialize field {name}")
                if name in res:                     # caller-owned buffer (out=): must already have this context's layout
                    shape = (B, _MAT_WIDTH[name]) if name in _MAT_WIDTH else ((4, _lib.NWLS) if name == "band_mean" else (B, self.nb))
                    pitch = self.row_pitch[_MAT_WIDTH[name]] if name in _MAT_WIDTH else shape[1]
                    t = res[name]
                    if not torch.is_tensor(t) or tuple(t.shape) != shape or t.dtype != td or t.device != self.device \
                            or (B > 1 and t.stride() != (pitch, 1)) or t.stride(-1) != 1:
                        raise ValueError(f"out[{name!r}] must be a {shape} {td} tensor with row stride {pitch} on {self.device}")
                elif name in _MAT_WIDTH:
                    res[name] = self._alloc_spec(B, _MAT_WIDTH[name], td)
                else:
                    res[name] = torch.empty((4, _lib.NWLS) if name == "band_mean" else (B, self.nb), dtype=td,
                                            device=self.device)
                setattr(mat, name, res[name].data_ptr())
        if _defer and _workspace is None:                  # a prepared call owns its scratch (other calls on the engine do not disturb it)
            n = int(self.lib.spart_workspace_bytes(self.ctx, dt, B))
            _workspace = torch.empty(max(n, 256), dtype=torch.uint8, device=self.device)
        if _workspace is not None:
            ws, wsn = ctypes.c_void_p(_workspace.data_ptr()), ctypes.c_size_t(_workspace.numel())
        else:
            ws, wsn = self._workspace(dt, B)
        args = (self.ctx, dt, B, col_ptrs if col_ptrs is not None else self._ptrs(cols),
                th[0].data_ptr() if th[0] is not None else None, th[1].data_ptr() if th[1] is not None else None,
                res["R_TOC"].data_ptr(), res["R_TOA"].data_ptr(), res["L_TOA"].data_ptr(),
                ctypes.byref(mat) if mat is not None else None, ws, wsn)
        if not _defer:
            self.calls["spart_run_batch"] += 1
            rc = self.lib.spart_run_batch(*args, self._stream())
            _lib.check(self.lib, self.ctx, rc)
            return res
        keep = (params, cols, th, li, rd, mat, res, _workspace)        # everything the argument pointers point into
        lib, ctx, calls, stream_of, device = self.lib, self.ctx, self.calls, self.torch.cuda.current_stream, self.device

        def call():
            calls["spart_run_batch"] += 1
            rc = lib.spart_run_batch(*args, ctypes.c_void_p(stream_of(device).cuda_stream))
            if rc:
                _lib.check(lib, ctx, rc)
            return keep[6]
        return call

    def prepare(self, params, dtype="float32", out=None, **kw):
        """run() with its argument marshalling done ONCE: returns ``call()`` which issues the same spart_run_batch on the current
        stream over the SAME resident buffers (``params`` a (27, B) float64 device tensor, ``out`` the preallocated results, any
        thermal / lidf tensors), with nothing but the ctypes call on the hot path -- what a loop over small batches wants when
        a HIP-graph capture is too rigid (the stream may change from call to call).  The call owns its workspace."""
        torch = self.torch
        if not (torch.is_tensor(params) and params.dim() == 2 and params.dtype == torch.float64 and params.is_contiguous()
                and params.device == self.device):
            raise ValueError("prepare() needs a contiguous (27, B) float64 tensor on the engine's device")
        if out is None or any(k not in out for k in ("R_TOC", "R_TOA", "L_TOA")):
            raise ValueError("prepare() needs preallocated out['R_TOC'|'R_TOA'|'L_TOA']")
        if kw.get("rdry") is not None:
            raise ValueError("prepare(): user dry-soil spectra are re-laid out per call (row pitch): use run()")
        for k in ("rho_thermal", "tau_thermal", "canopy_lidf"):
            v = kw.get(k)
            if v is not None and not (torch.is_tensor(v) and v.device == self.device):
                raise ValueError(f"prepare(): {k} must be a tensor on the engine's device (its storage is reused by every call)")
        return self.run(params, dtype, out=out, _defer=True, **kw)

    def _check_out(self, name, t, shape, td):
        """a caller-supplied output must be exactly what the kernels write: they get its data_ptr() and nothing else"""
        if not self.torch.is_tensor(t) or tuple(t.shape) != tuple(shape) or t.dtype != td or t.device != self.device \
                or not t.is_contiguous():
            raise ValueError(f"out[{name!r}] must be a contiguous {tuple(shape)} {td} tensor on {self.device}, got "
                             f"{tuple(t.shape) if self.torch.is_tensor(t) else type(t)} "
                             f"{getattr(t, 'dtype', None)} on {getattr(t, 'device', None)}")

    def capture(self, params, dtype="float32", out=None, **kw):
        """Record one run() over RESIDENT buffers into a HIP graph and return its replay function (no arguments;
        results land in ``out``).  For steps of a few kernels on small batches (100k spectra: 1.3 ms) the replay
        removes the per-launch host work and the gaps between the dependent kernels.  ``params`` must be a (27, B)
        float64 device tensor and ``out`` the three (B, nb) result tensors; the graph owns its workspace, so other
        calls on this engine do not disturb it.  spart_run_batch allocates nothing and never synchronises, which is
        what makes it capturable."""
        torch = self.torch
        if not (torch.is_tensor(params) and params.dim() == 2 and params.dtype == torch.float64 and params.is_contiguous()
                and params.device == self.device):
            raise ValueError("capture() needs a contiguous (27, B) float64 tensor on the engine's device")
        if out is None or any(k not in out for k in ("R_TOC", "R_TOA", "L_TOA")):
            raise ValueError("capture() needs preallocated out['R_TOC'|'R_TOA'|'L_TOA']")
        dt = DTYPES[dtype]
        n = int(self.lib.spart_workspace_bytes(self.ctx, dt, params.shape[1]))
        ws = torch.empty(max(n, 256), dtype=torch.uint8, device=self.device)
        s = torch.cuda.Stream(self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            self.run(params, dtype, out=out, _workspace=ws, **kw)         # warm-up outside the capture
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                res = self.run(params, dtype, out=out, _workspace=ws, **kw)
        torch.cuda.current_stream(self.device).wait_stream(s)
        keep = (ws, params, res)                                           # buffers the graph points into

        def replay():
            g.replay()
            return keep[2]
        replay.graph = g
        return replay

    def lut_nearest(self, lut, obs, weights=None, dtype="float32", stats=False):
        """LUT inversion: for each row of obs (M, nb) the index of THE closest row of lut (B, nb) under the weighted squared
        distance ``c = sum_j (w_j * d_j) * d_j``, ``d = lut - obs`` (sequential, rounded to ``dtype``, no FMA), lowest index on
        ties, and that distance -- bit-exact against a brute-force loop (include/spart_hip.h: spart_lut_nearest).
        -> (idx (M,) int64 tensor, cost (M,) tensor); with ``stats=True`` also a dict with the number of observations that
        took the brute-force path and the scale Nmax of the rounding bound (spart_lut_stats; synchronises)."""
        torch = self.torch
        dt = DTYPES[dtype]
        td = self._tdtype(dt)
        lut = torch.as_tensor(lut).to(device=self.device, dtype=td).contiguous()
        obs = torch.as_tensor(obs).to(device=self.device, dtype=td).contiguous()
        if lut.dim() != 2 or obs.dim() != 2 or lut.shape[1] != obs.shape[1]:
            raise ValueError("lut (B, nb) and obs (M, nb) must share nb")
        w = None if weights is None else torch.as_tensor(weights).to(device=self.device, dtype=td).contiguous()
        if w is not None and w.numel() != lut.shape[1]:
            raise ValueError(f"weights has {w.numel()} entries, expected nb = {lut.shape[1]}")
        B, nb = lut.shape
        M = obs.shape[0]
        idx = torch.empty((M,), dtype=torch.int64, device=self.device)
        cost = torch.empty((M,), dtype=td, device=self.device)
        n = int(self.lib.spart_lut_workspace_bytes(dt, B, nb, M))
        ws = torch.empty(max(n, 256), dtype=torch.uint8, device=self.device)
        self.calls["spart_lut_nearest"] += 1
        rc = self.lib.spart_lut_nearest(self.ctx, dt, B, nb, lut.data_ptr(), M, obs.data_ptr(),
                                        w.data_ptr() if w is not None else None, idx.data_ptr(), cost.data_ptr(),
                                        ws.data_ptr(), ctypes.c_size_t(ws.numel()), self._stream())
        _lib.check(self.lib, self.ctx, rc)
        if not stats:
            return idx, cost
        nbf, nmax = ctypes.c_int64(0), ctypes.c_double(0.0)
        if M > 0:
            torch.cuda.current_stream(self.device).synchronize()
            _lib.check(self.lib, self.ctx, self.lib.spart_lut_stats(self.ctx, dt, B, nb, M, ws.data_ptr(), ctypes.byref(nbf),
                                                                    ctypes.byref(nmax)))
        return idx, cost, {"brute_force": int(nbf.value), "nmax": float(nmax.value)}

    def profile(self, max_calls):
        """bracket the band kernel of the next ``max_calls`` run() calls with HIP events (0 = off)."""
        _lib.check(self.lib, self.ctx, self.lib.spart_profile_enable(self.ctx, int(max_calls)))

    def profile_read(self):
        """-> (summed band-kernel milliseconds, number of timed calls)"""
        ms, n = ctypes.c_double(0.0), ctypes.c_int(0)
        _lib.check(self.lib, self.ctx, self.lib.spart_profile_read(self.ctx, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def profile_read_stages(self):
        """-> ({'prelude': ms, 'bands': ms, 'columns': ms} summed over the timed calls, number of calls)"""
        ms, n = (ctypes.c_double * 8)(), ctypes.c_int(0)      # (room for builds of earlier rounds that report four stages)
        _lib.check(self.lib, self.ctx, self.lib.spart_profile_read_stages(self.ctx, ms, ctypes.byref(n)))
        return dict(zip(_lib.STAGES, [float(x) for x in ms])), n.value

    def econv(self):
        out = np.zeros(self.nb)
        _lib.check(self.lib, self.ctx, self.lib.spart_ctx_econv(self.ctx, _dp(out)))
        return out


_cache_lock = threading.RLock()   # guards the three caches below (contexts are thread-safe; so is finding one)
_engines = {}            # (sensor name, device) -> Engine built from the packaged tables: no hashing on this path
_by_content = {}         # (table digest, sensor digest | None, device) -> Engine, least recently used last out
_packaged_digest = {}    # memo: digests of the packaged tables ("optical") and sensors (name)
MAX_CONTENT_ENGINES = 16  # a context is ~0.6 MB of device tables; a handful of table sets at most is expected


def get_engine(sensor=None, device=None, optical_params=None, et_params=None, sensor_info=None, need=OPTICAL_KEYS):
    """The engine (device context) for a sensor and a set of tables.

    get_engine(sensor, device): the packaged tables and the packaged sensor ``sensor`` (or no sensor), one engine per
    (name, device), found without looking at any table.
    With ``optical_params`` / ``et_params`` / ``sensor_info`` (the reference's dicts: SPART.optipar, SPART.ETpar,
    SPART.sensorinfo; PROSPECT_5D's and BSM's second argument): the engine whose device tables have exactly THAT content
    -- the key is a digest of the arrays, re-computed on every call (~20 us with xxhash, 0.2 ms with sha1), so editing
    a dict or one of its arrays in place between two calls gives the second call the edited tables, as in the
    reference, and handing in the unmodified packaged dicts gives the very engine of the name-keyed path.
    ``need``: the keys of optical_params the calling entry point reads (see optical_block)."""
    torch = _require_gpu()
    if device is None:
        device = torch.cuda.current_device()
    device = int(device)
    with _cache_lock:
        return _get_engine_locked(sensor, device, optical_params, et_params, sensor_info, need)


def _get_engine_locked(sensor, device, optical_params, et_params, sensor_info, need):
    if optical_params is None and et_params is None and sensor_info is None:
        key = (sensor, device)
        if key not in _engines:
            _engines[key] = Engine(sensor, device)
        return _engines[key]
    if "optical" not in _packaged_digest:
        _packaged_digest["optical"] = _digest(optical_block().values())
    ob = optical_block(optical_params, et_params, need)
    dt = _digest(ob.values())
    ds = None
    if sensor_info is not None:
        ds = _digest(sensor_block(sensor_info).values())
    elif sensor is not None:
        ds = _packaged_digest.get(sensor)
        if ds is None:
            ds = _packaged_digest[sensor] = _digest(sensor_block(tables.load_sensor_info(sensor)).values())
    # the packaged content under whatever dict it arrives in IS the name-keyed engine
    if dt == _packaged_digest["optical"]:
        if ds is None:
            return get_engine(None, device)
        for name in ([sensor] if isinstance(sensor, str) else []) + list(tables.SENSORS):
            if name not in _packaged_digest:
                try:
                    _packaged_digest[name] = _digest(sensor_block(tables.load_sensor_info(name)).values())
                except FileNotFoundError:
                    continue
            if _packaged_digest[name] == ds:
                return get_engine(name, device)
    key = (dt, ds, device)
    eng = _by_content.pop(key, None)
    if eng is None:
        while len(_by_content) >= MAX_CONTENT_ENGINES:
            _by_content.pop(next(iter(_by_content)))
        eng = Engine(sensor if sensor_info is None else None, device, sensor_info=sensor_info, optical_params=optical_params,
                     et_params=et_params, need=need)
    _by_content[key] = eng                    # (re-inserted at the end: most recently used)
    return eng
