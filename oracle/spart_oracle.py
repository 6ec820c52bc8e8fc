"""CPU oracle for the SPART hot path  --  TEST INFRASTRUCTURE, NOT THE PRODUCT.

A numpy (float64) restatement of the reference algorithm
``parameter vector -> BSM + PROSPECT-5D/PRO + SAILH + SMAC -> R_TOC / R_TOA / L_TOA``
of wirrell/SPART-python, vectorised over a batch axis.  Every function cites the reference
file:line it follows.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import this module; the shipped package (``spart-python_amd/``)
never does and fails loudly when its HIP library is missing.

Parity status: PINNED.  ``tests/golden/*.npz`` hold outputs of the real reference, imported
in the build container (``tests/golden/make_golden.py``); ``tests/test_oracle_golden.py``
checks this restatement against them (the reference's own golden parquet files are absent
from the snapshot, SURVEY.md §4, and its e2e test only asserts positivity).

Third-party arithmetic on the path (versions unpinned upstream; the goldens were made with
numpy 2.2.6 / scipy 1.15.3):
  * ``scipy.integrate.quad`` for E1(K) (prospect_5d.py:185-194).  Restated with the
    closed form ``scipy.special.exp1`` (agrees to ~1e-10 rel); ``e1="quad"`` reproduces
    the reference call literally.
  * ``scipy.integrate.quad`` for the 61 hot-spot layer integrals (sailh.py:131-135).
    ``pso="quad"`` reproduces the call; ``pso="gl"`` uses graded Gauss-Legendre panels
    (vectorised, used for the CPU baseline timing).
  * ``scipy.stats.poisson.pmf`` (bsm.py:121) restated as exp(-mu) mu^k / k!.

Layout of the parameter matrix ``P`` (B, 27), SURVEY.md §8(d):
  leaf   0..8   Cab, Cdm, Cw, Cs, Cca, Cant, N, PROT, CBC
  soil   9..14  B, lat, lon, SMp, SMC, film
  canopy 15..18 LAI, LIDFa, LIDFb, q
  angles 19..21 tts, tto, psi
  atm    22..25 aot550, uo3, uh2o, Pa
  DOY    26
"""
from __future__ import annotations

import math
import os

import numpy as np

NWL = 2001           # 400..2400 nm                          (SPART.py:303)
NWLT = 161           # thermal pad 2500..15000/100, 16000..50000/1000 (SPART.py:307-309)
NWLS = NWL + NWLT    # 2162                                  (SPART.py:310)
NL = 60              # canopy layers                         (sailh.py:345)

COEF_NAMES = [
    "ah2o", "nh2o", "ao3", "no3", "ao2", "no2", "po2", "aco2", "nco2", "pco2",
    "ach4", "nch4", "pch4", "ano2", "nno2", "pno2", "aco", "nco", "pco",
    "a0s", "a1s", "a2s", "a3s", "a0T", "a1T", "a2T", "a3T", "taur",
    "a0taup", "a1taup", "wo", "gc", "a0P", "a1P", "a2P", "a3P", "a4P",
    "Rest1", "Rest2", "Rest3", "Rest4", "Resr1", "Resr2", "Resr3",
    "Resa1", "Resa2", "Resa3", "Resa4",
]

_DEFAULT_TABLES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                               "spart-python_amd", "spart_amd", "data", "spart_tables.npz")


def wl_solar():
    """SpectralBands.wlS (SPART.py:303-310)."""
    return np.concatenate([np.arange(400, 2401, 1), np.arange(2500, 15001, 100),
                           np.arange(16000, 50001, 1000)]).astype(np.float64)


def load_tables(path=None):
    """The exported tables (tools/export_tables.py); plain arrays, no pickle."""
    z = np.load(path or _DEFAULT_TABLES)
    return {k: z[k] for k in z.files}


def sensor_tables(tables, sensor):
    if f"{sensor}/wl_smac" not in tables:
        # same failure class as SPART.py:421-423 (open() of a missing pickle)
        raise FileNotFoundError(f"sensor_information/{sensor}.pkl")
    return {k.split("/", 1)[1]: v for k, v in tables.items() if k.startswith(sensor + "/")}


# --------------------------------------------------------------------------- PROSPECT
def calculate_tav(alpha, nr):
    """Stern/Allen average interface transmissivity, prospect_5d.py:249-311."""
    rd = np.pi / 180
    n2 = nr ** 2
    n_p = n2 + 1
    nm = n2 - 1
    a = (nr + 1) * (nr + 1) / 2
    k = -(n2 - 1) * (n2 - 1) / 4
    sa = np.sin(alpha * rd)
    b1 = 0
    if alpha != 90:
        b1 = np.sqrt((sa ** 2 - n_p / 2) * (sa ** 2 - n_p / 2) + k)
    b2 = sa ** 2 - n_p / 2
    b = b1 - b2
    b3 = b ** 3
    a3 = a ** 3
    ts = (k ** 2 / (6 * b3) + k / b - b / 2) - (k ** 2 / (6 * a3) + k / a - a / 2)
    tp1 = -2 * n2 * (b - a) / (n_p ** 2)
    tp2 = -2 * n2 * n_p * np.log(b / a) / (nm ** 2)
    tp3 = n2 * (1 / b - 1 / a) / 2
    tp4 = (16 * n2 ** 2 * (n2 ** 2 + 1)
           * np.log((2 * n_p * b - nm ** 2) / (2 * n_p * a - nm ** 2)) / (n_p ** 3 * nm ** 2))
    tp5 = 16 * n2 ** 3 * (1 / (2 * n_p * b - nm ** 2) - 1 / (2 * n_p * a - nm ** 2)) / n_p ** 3
    tp = tp1 + tp2 + tp3 + tp4 + tp5
    return (ts + tp) / (2 * sa ** 2)


def _expint_quad(K):
    """Literal restatement of prospect_5d.py:185-194 (scipy QUADPACK qagie per element)."""
    import scipy.integrate as integrate

    out = np.empty_like(K)
    flat, o = K.ravel(), out.ravel()
    for i, x in enumerate(flat):
        o[i] = integrate.quad(lambda t: np.exp(-t) / t, x, np.inf)[0]
    return out


def prospect_5d(leaf, tables, e1="exp1"):
    """PROSPECT-5D / PROSPECT-PRO, prospect_5d.py:117-246.

    leaf : (B, 9) [Cab, Cdm, Cw, Cs, Cca, Cant, N, PROT, CBC]
    returns refl, tran, kChlrel each (B, 2001)
    """
    from scipy.special import exp1

    leaf = np.atleast_2d(np.asarray(leaf, dtype=np.float64))
    col = lambda i: leaf[:, i:i + 1]
    Cab, Cdm, Cw, Cs, Cca, Cant, N, PROT, CBC = (col(i) for i in range(9))
    # PROSPECT-PRO rule (prospect_5d.py:148-155): Cdm forced to 0
    pro = ((PROT > 0.0) | (CBC > 0.0)) & (Cdm > 0)
    Cdm = np.where(pro, 0.0, Cdm)

    t = tables
    nr = t["nr"][None, :]
    with np.errstate(all="ignore"):
        # prospect_5d.py:170-179
        Kall = (Cab * t["Kab"] + Cca * t["Kca"] + Cdm * t["Kdm"] + Cw * t["Kw"] + Cs * t["Ks"]
                + Cant * t["Kant"] + CBC * t["cbc"] + PROT * t["prot"]) / N
        pos = Kall > 0                                      # :182
        t1 = (1 - Kall) * np.exp(-Kall)                     # :183
        Ksafe = np.where(pos, Kall, 1.0)
        E1 = _expint_quad(Ksafe) if e1 == "quad" else exp1(Ksafe)
        t2 = Kall ** 2 * E1                                 # :194
        tau = np.where(pos, t1 + t2, 1.0)                   # :195-196
        kChlrel = np.where(pos, Cab * t["Kab"] / (Ksafe * N), 0.0)  # :197-198

        t_alph = calculate_tav(40, nr)                      # :200-205
        r_alph = 1 - t_alph
        t12 = calculate_tav(90, nr)
        r12 = 1 - t12
        t21 = t12 / (nr ** 2)
        r21 = 1 - t21

        denom = 1 - r21 * r21 * tau ** 2                    # :208-214
        Ta = t_alph * tau * t21 / denom
        Ra = r_alph + r21 * tau * Ta
        tt = t12 * tau * t21 / denom
        r = r12 + r21 * tau * tt

        D = np.sqrt((1 + r + tt) * (1 + r - tt) * (1 - r + tt) * (1 - r - tt))  # :219
        rq = r ** 2
        tq = tt ** 2
        a = (1 + rq - tq + D) / (2 * r)
        b = (1 - rq + tq + D) / (2 * tt)
        bNm1 = b ** (N - 1)                                 # :225-230
        bN2 = bNm1 ** 2
        a2 = a ** 2
        denom = a2 * bN2 - 1
        Rsub = a * (bN2 - 1) / denom
        Tsub = bNm1 * (a2 - 1) / denom

        j = (r + tt) >= 1                                   # :233-235
        Tsub_j = tt / (tt + (1 - tt) * (N - 1))
        Tsub = np.where(j, Tsub_j, Tsub)
        Rsub = np.where(j, 1 - Tsub_j, Rsub)

        denom = 1 - Rsub * r                                # :239-241
        tran = Ta * Tsub / denom
        refl = Ra + Ta * Rsub * tt / denom
    return refl, tran, kChlrel


# --------------------------------------------------------------------------- BSM
def poisson_pmf(k, mu):
    """scipy.stats.poisson.pmf(k, mu) for small integer k (bsm.py:121)."""
    k = np.asarray(k)
    fact = np.array([math.factorial(int(i)) for i in k.ravel()], dtype=np.float64).reshape(k.shape)
    return np.exp(-mu) * mu ** k / fact


def bsm(soil, tables, rdry=None):
    """BSM + soilwat, bsm.py:17-59, 62-128.

    soil : (B, 6) [B, lat, lon, SMp, SMC, film];  rdry : optional (B, 2001) user spectra (bsm.py:42-43)
    returns refl (wet), refl_dry each (B, 2001)
    """
    soil = np.atleast_2d(np.asarray(soil, dtype=np.float64))
    col = lambda i: soil[:, i:i + 1]
    Bb, lat, lon, SMp, SMC, film = (col(i) for i in range(6))
    if rdry is None:
        GSV = tables["GSV"]
        f1 = Bb * np.sin(lat * np.pi / 180)                                   # bsm.py:49-52
        f2 = Bb * np.cos(lat * np.pi / 180) * np.sin(lon * np.pi / 180)
        f3 = Bb * np.cos(lat * np.pi / 180) * np.cos(lon * np.pi / 180)
        rdry = f1 * GSV[None, :, 0] + f2 * GSV[None, :, 1] + f3 * GSV[None, :, 2]
    else:
        rdry = np.atleast_2d(np.asarray(rdry, dtype=np.float64))
    kw = tables["Kw"][None, :]
    nw = tables["nw"][None, :]

    mu = (SMp - 5) / SMC                                                      # :101
    # :110-119 (table-only quantities)
    c_bac = calculate_tav(90, 2 / nw) / calculate_tav(90, 2)
    rbac = 1 - (1 - rdry) * (rdry * c_bac + 1 - rdry)
    p = 1 - calculate_tav(90, nw) / nw ** 2
    Rw = 1 - calculate_tav(40, nw)
    k = np.arange(7)
    with np.errstate(all="ignore"):
        fmul = poisson_pmf(k[None, :], np.where(mu > 0, mu, 1.0))            # (B,7)  :121
        rwet = rdry * fmul[:, 0:1]
        for kk in range(1, 7):
            tw = np.exp(-2 * kw * film * kk)                                 # :122
            Rwet_k = Rw + (1 - Rw) * (1 - p) * tw * rbac / (1 - p * tw * rbac)  # :123
            rwet = rwet + Rwet_k * fmul[:, kk:kk + 1]                        # :124
    rwet = np.where(mu <= 0, rdry, rwet)                                     # :102-103
    return rwet, rdry


# --------------------------------------------------------------------------- SAILH
def calculate_leafangles(LIDFa, LIDFb):
    """LIDF fixed-point iteration, sailh.py:351-398.  returns (B, 13)."""
    a = np.atleast_1d(np.asarray(LIDFa, dtype=np.float64))
    b = np.atleast_1d(np.asarray(LIDFb, dtype=np.float64))
    thetas = [10.0 * i for i in range(1, 9)] + [80.0 + 2.0 * i for i in range(1, 5)]
    F = np.zeros((a.shape[0], 14))
    rd = np.pi / 180
    for i, theta in enumerate(thetas, start=1):
        # dcum, sailh.py:368-384
        x = np.full_like(a, 2 * rd * theta)
        theta2 = 2 * rd * theta
        y = np.zeros_like(a)
        active = np.ones_like(a, dtype=bool)
        while active.any():
            yn = a * np.sin(x) + 0.5 * b * np.sin(2 * x)
            dx = 0.5 * (yn - x + theta2)
            y = np.where(active, yn, y)
            x = np.where(active, x + dx, x)
            active = active & (np.abs(dx) > 1e-8)
        f = (2 * y + theta2) / np.pi
        f = np.where(a > 1, 1 - np.cos(theta * rd), f)      # sailh.py:371-372
        F[:, i] = f
    F[:, 13] = 1
    return np.diff(F, axis=1)


def volscatt(sin_tts, cos_tts, sin_tto, cos_tto, psi_rad, sin_ttli, cos_ttli):
    """sailh.py:401-446; all sample args (B,1), leaf-angle args (1,13)."""
    cos_psi = np.cos(psi_rad)
    Cs = cos_ttli * cos_tts
    Ss = sin_ttli * sin_tts
    Co = cos_ttli * cos_tto
    So = sin_ttli * sin_tto
    As = np.maximum(Ss, Cs)
    Ao = np.maximum(So, Co)
    bts = np.arccos(-Cs / As)
    bto = np.arccos(-Co / Ao)
    chi_o = 2 / np.pi * ((bto - np.pi / 2) * Co + np.sin(bto) * So)
    chi_s = 2 / np.pi * ((bts - np.pi / 2) * Cs + np.sin(bts) * Ss)
    delta1 = np.abs(bts - bto)
    delta2 = np.pi - np.abs(bts + bto - np.pi)
    Tot = psi_rad + delta1 + delta2
    bt1 = np.minimum(psi_rad, delta1)
    bt3 = np.maximum(psi_rad, delta2)
    bt2 = Tot - bt1 - bt3
    T1 = 2 * Cs * Co + Ss * So * cos_psi
    T2 = np.sin(bt2) * (2 * As * Ao + Ss * So * np.cos(bt1) * np.cos(bt3))
    Jmin = bt2 * T1 - T2
    Jplus = (np.pi - bt2) * T1 + T2
    frho = np.maximum(0.0, Jplus / (2 * np.pi ** 2))
    ftau = np.maximum(0.0, -Jmin / (2 * np.pi ** 2))
    return chi_s, chi_o, frho, ftau


def _pso_function(x, K, k, LAI, q, dso):
    """sailh.py:118-129 (scalar)."""
    if dso != 0:
        alpha = (dso / q) * 2 / (k + K)
        return np.exp((K + k) * LAI * x + np.sqrt(K * k) * LAI / alpha * (1 - np.exp(x * alpha)))
    return np.exp((K + k) * LAI * x - np.sqrt(K * k) * LAI * x)


def _pso_quad(K, k, LAI, q, dso, nl=NL):
    """nl + 1 layer integrals by scipy quad, literally sailh.py:131-135 (nl = canopy.nlayers, :48).  returns (B, nl + 1)
    (np.arange may give xl one point more than nl + 1, :52; the reference only reads Pso[0:nl] and Pso[nl], :216, 219)."""
    import scipy.integrate as integrate

    B = K.shape[0]
    dx = 1 / nl
    xl = np.arange(0, -1 - (1 / nl), -1 / nl)       # sailh.py:52
    Pso = np.zeros((B, nl + 1))
    for i in range(B):
        args = (float(K[i]), float(k[i]), float(LAI[i]), float(q[i]), float(dso[i]))
        for j in range(nl + 1):
            Pso[i, j] = integrate.quad(_pso_function, xl[j] - dx, xl[j], args=args)[0] / dx
    return Pso


_GL_X, _GL_W = np.polynomial.legendre.leggauss(20)


def _pso_gl(K, k, LAI, q, dso, nl=NL):
    """Vectorised alternative: returns (sum_{j<nl} Pso_j, Pso_nl) from graded Gauss-Legendre panels.

    sum_j Pso_j * dx == integral over [-1, 0]; Pso_nl * dx == integral over [-1-dx, -1]
    (sailh.py:131-135, 216, 219).  Panels halve towards x = 0 where the integrand varies on
    the scale 1/alpha.
    """
    dx = 1 / nl
    A = (K + k) * LAI
    with np.errstate(all="ignore"):
        alpha = np.where(dso != 0, (dso / q) * 2 / (k + K), 1.0)
        C = np.where(dso != 0, np.sqrt(K * k) * LAI / alpha, 0.0)
        beta0 = np.sqrt(K * k) * LAI       # dso == 0 branch

    def f(x):
        with np.errstate(all="ignore"):
            g1 = A * x + C * (-np.expm1(x * alpha))
            g0 = A * x - beta0 * x
            return np.exp(np.where(dso != 0, g1, g0))

    def panel(a, b):
        # a, b: (B,) or scalars
        a = np.broadcast_to(np.asarray(a, dtype=np.float64), K.shape)
        b = np.broadcast_to(np.asarray(b, dtype=np.float64), K.shape)
        h = 0.5 * (b - a)
        c = 0.5 * (b + a)
        tot = np.zeros_like(K)
        for xi, wi in zip(_GL_X, _GL_W):
            tot = tot + wi * f(c + h * xi)
        return tot * h

    nlev = 24
    total = panel(-1.0, -0.5)
    lo = -0.5
    for _ in range(nlev):
        total = total + panel(lo, lo / 2)
        lo = lo / 2
    total = total + panel(lo, 0.0)
    npan = max(2, 2 * int(math.ceil(NL / nl)))      # (panels no wider than 1/120, as for the default 60 layers)
    h = dx / npan
    below = sum(panel(-1.0 - (i + 1) * h, -1.0 - i * h) for i in range(npan))
    return total / dx, below / dx


def sailh(rho, tau, rs, canopy, angles, pso="quad", lidf=None, nlayers=NL):
    """SAILH, sailh.py:14-237.

    rho, tau, rs : (B, 2162) leaf reflectance / transmittance, soil reflectance (thermal padded)
    canopy : (B, 4) [LAI, LIDFa, LIDFb, q];  angles : (B, 3) [tts, tto, psi]
    lidf : canopy.lidf as SAILH reads it from the object (sailh.py:51), (13,), (13, 1) or (B, 13); None = the constructor's
           calculate_leafangles(LIDFa, LIDFb) (sailh.py:348).  nlayers : canopy.nlayers (sailh.py:48)
    returns dict rso, rdo, rsd, rdd (B, 2162) and the sample scalars in 'aux'
    """
    NL = int(nlayers)                                   # (shadows the module default below: sailh.py:48, 52-54)
    rho = np.atleast_2d(rho)
    tau = np.atleast_2d(tau)
    rs = np.atleast_2d(rs)
    if rho.shape[1] != NWLS:   # sailh.py:37-44
        raise RuntimeError("Parameter leafopt.refl must be of len 2162 i.e. include thermal specturm.")
    canopy = np.atleast_2d(np.asarray(canopy, dtype=np.float64))
    angles = np.atleast_2d(np.asarray(angles, dtype=np.float64))
    LAI = canopy[:, 0:1]
    q = canopy[:, 3:4]
    if lidf is None:
        lidf = calculate_leafangles(canopy[:, 1], canopy[:, 2])     # sailh.py:348
    else:
        lidf = np.asarray(lidf, dtype=np.float64)
        lidf = lidf.reshape(1, 13) if lidf.size == 13 else lidf       # (13,) / (13, 1) -> one row, broadcast over the batch
    deg2rad = np.pi / 180
    litab = np.array([*range(5, 80, 10), *range(81, 91, 2)], dtype=np.float64)[None, :]   # :49
    dx = 1 / NL
    iLAI = LAI * dx
    tts = angles[:, 0:1]
    tto = angles[:, 1:2]
    rel = angles[:, 2:3]
    psi = np.abs(rel - 360 * np.round(rel / 360))                    # :65 (round-half-even, as Python)
    psi_rad = psi * deg2rad
    sin_tts, cos_tts, tan_tts = np.sin(tts * deg2rad), np.cos(tts * deg2rad), np.tan(tts * deg2rad)
    sin_tto, cos_tto, tan_tto = np.sin(tto * deg2rad), np.cos(tto * deg2rad), np.tan(tto * deg2rad)
    sin_ttli, cos_ttli = np.sin(litab * deg2rad), np.cos(litab * deg2rad)
    with np.errstate(all="ignore"):
        dso = np.sqrt(tan_tts ** 2 + tan_tto ** 2 - 2 * tan_tts * tan_tto * np.cos(psi_rad))   # :78
        chi_s, chi_o, frho, ftau = volscatt(sin_tts, cos_tts, sin_tto, cos_tto, psi_rad, sin_ttli, cos_ttli)
        ksli = chi_s / cos_tts                                        # :85-90
        koli = chi_o / cos_tto
        sobli = frho * np.pi / (cos_tts * cos_tto)
        sofli = ftau * np.pi / (cos_tts * cos_tto)
        bfli = cos_ttli ** 2
        dot = lambda m: np.sum(m * lidf, axis=1, keepdims=True)      # :93-97
        k, K, bf, sob, sof = dot(ksli), dot(koli), dot(bfli * np.ones_like(ksli)), dot(sobli), dot(sofli)
        sdb, sdf = 0.5 * (k + bf), 0.5 * (k - bf)                     # :100-105
        ddb, ddf = 0.5 * (1 + bf), 0.5 * (1 - bf)
        dob, dof = 0.5 * (K + bf), 0.5 * (K - bf)

        # hot spot, :115-135 (Ps/Po of :108-113 are dead code)
        if pso == "quad":
            Pso = _pso_quad(K[:, 0], k[:, 0], LAI[:, 0], q[:, 0], dso[:, 0], NL)
            sumPso = np.sum(Pso[:, 0:NL], axis=1, keepdims=True)
            Pso2w = Pso[:, NL:NL + 1]
        else:
            s, b2 = _pso_gl(K[:, 0], k[:, 0], LAI[:, 0], q[:, 0], dso[:, 0], NL)
            sumPso, Pso2w = s[:, None], b2[:, None]

        sigb = ddb * rho + ddf * tau                                   # :142-152
        sigf = ddf * rho + ddb * tau
        sb = sdb * rho + sdf * tau
        sf = sdf * rho + sdb * tau
        vb = dob * rho + dof * tau
        vf = dof * rho + dob * tau
        w = sob * rho + sof * tau
        a = 1 - sigf
        m = np.sqrt(a ** 2 - sigb ** 2)
        rinf = (a - m) / sigb
        rinf2 = rinf * rinf

        def calcJ1(x, m, kk, LAI):                                     # :154-170
            sing = np.abs((m - kk) * LAI) < 1e-6
            normal = (np.exp(m * LAI * x) - np.exp(kk * LAI * x)) / (kk - m)
            singular = (-0.5 * (np.exp(m * LAI * x) + np.exp(kk * LAI * x)) * LAI * x
                        * (1 - 1 / 12 * (kk - m) ** 2 * LAI ** 2))
            return np.where(sing, singular, normal)

        def calcJ2(x, m, kk, LAI):                                     # :172-177
            return (np.exp(kk * LAI * x) - np.exp(-kk * LAI) * np.exp(-m * LAI * (1 + x))) / (kk + m)

        J1k = calcJ1(-1, m, k, LAI)                                    # :180-183
        J2k = calcJ2(0, m, k, LAI)
        J1K = calcJ1(-1, m, K, LAI)
        J2K = calcJ2(0, m, K, LAI)
        e1 = np.exp(-m * LAI)                                          # :185-198
        e2 = e1 ** 2
        re = rinf * e1
        denom = 1 - rinf2 ** 2
        s1 = sf + rinf * sb
        s2 = sf * rinf + sb
        v1 = vf + rinf * vb
        v2 = vf * rinf + vb
        Pss = s1 * J1k
        Qss = s2 * J2k
        Poo = v1 * J1K
        Qoo = v2 * J2K
        tau_ss = np.exp(-k * LAI)                                      # :200-210
        tau_oo = np.exp(-K * LAI)
        Z = (1 - tau_ss * tau_oo) / (K + k)
        tau_dd = (1 - rinf2) * e1 / denom
        rho_dd = rinf * (1 - e2) / denom
        tau_sd = (Pss - re * Qss) / denom
        tau_do = (Poo - re * Qoo) / denom
        rho_sd = (Qss - re * Pss) / denom
        rho_do = (Qoo - re * Poo) / denom
        T1 = v2 * s1 * (Z - J1k * tau_oo) / (K + m) + v1 * s2 * (Z - J1K * tau_ss) / (k + m)   # :212-214
        T2 = -(Qoo * rho_sd + Poo * tau_sd) * rinf
        rho_sod = (T1 + T2) / (1 - rinf2)
        rho_sos = w * sumPso * iLAI                                    # :216-219
        rho_so = rho_sod + rho_sos
        denom = 1 - rs * rho_dd                                        # :222-233
        rso = (rho_so + rs * Pso2w
               + ((tau_sd + tau_ss * rs * rho_dd) * tau_oo + (tau_sd + tau_ss) * tau_do) * rs / denom)
        rdo = rho_do + (tau_oo + tau_do) * rs * tau_dd / denom
        rsd = rho_sd + (tau_ss + tau_sd) * rs * tau_dd / denom
        rdd = rho_dd + tau_dd * rs * tau_dd / denom
    aux = dict(lidf=lidf, k=k, K=K, bf=bf, sob=sob, sof=sof, dso=dso, sumPso=sumPso, Pso2w=Pso2w,
               tau_ss=tau_ss, tau_oo=tau_oo)
    return dict(rso=rso, rdo=rdo, rsd=rsd, rdd=rdd, aux=aux)


# --------------------------------------------------------------------------- SMAC
SMAC_OUT = ["Ta_s", "Ta_o", "Tg", "Ra_dd", "Ra_so", "Ta_ss", "Ta_sd", "Ta_oo", "Ta_do"]


def _smac_core(tts, tto, psi, Pa, taup550, uo3, uh2o, c):
    """smac.py:94-211 with whatever operand types it is handed (scalars or (B,1) arrays)."""
    cdr = np.pi / 180
    crd = 180 / np.pi
    us = np.cos(tts * cdr)
    uv = np.cos(tto * cdr)
    Peq = Pa / 1013.25
    m = 1 / us + 1 / uv
    taup = c["a0taup"] + c["a1taup"] * taup550
    uo2 = Peq ** c["po2"]
    uco2 = Peq ** c["pco2"]
    uch4 = Peq ** c["pch4"]
    uno2 = Peq ** c["pno2"]
    uco = Peq ** c["pco"]
    to3 = np.exp(c["ao3"] * (uo3 * m) ** c["no3"])
    th2o = np.exp(c["ah2o"] * (uh2o * m) ** c["nh2o"])
    to2 = np.exp(c["ao2"] * (uo2 * m) ** c["no2"])
    tco2 = np.exp(c["aco2"] * (uco2 * m) ** c["nco2"])
    tch4 = np.exp(c["ach4"] * (uch4 * m) ** c["nch4"])
    tno2 = np.exp(c["ano2"] * (uno2 * m) ** c["nno2"])
    tco = np.exp(c["aco"] * (uco * m) ** c["nco"])
    tg = th2o * to3 * to2 * tco2 * tch4 * tco * tno2
    s = c["a0s"] * Peq + c["a3s"] + c["a1s"] * taup550 + c["a2s"] * taup550 ** 2
    ttetas = c["a0T"] + c["a1T"] * taup550 / us + (c["a2T"] * Peq + c["a3T"]) / (1 + us)
    ttetav = c["a0T"] + c["a1T"] * taup550 / uv + (c["a2T"] * Peq + c["a3T"]) / (1 + uv)
    # NB the reference multiplies degrees by 180/pi (smac.py:130); reproduced on purpose
    cksi = -((us * uv) + (np.sqrt(1 - us * us) * np.sqrt(1 - uv * uv) * np.cos(psi * crd)))
    cksi = np.where(cksi < -1, -1.0, cksi)                      # smac.py:134-135
    ksiD = crd * np.arccos(cksi)
    ray_phase = 0.7190443 * (1 + (cksi * cksi)) + 0.0412742
    ray_ref = (c["taur"] * ray_phase) / (4 * us * uv)
    ray_ref = ray_ref * Pa / 1013.25
    taurz = c["taur"] * Peq
    aer_phase = c["a0P"] + c["a1P"] * ksiD + c["a2P"] * ksiD * ksiD + c["a3P"] * ksiD ** 3 + c["a4P"] * ksiD ** 4
    wo, gc = c["wo"], c["gc"]
    ak2 = (1 - wo) * (3 - wo * 3 * gc)
    ak = np.sqrt(ak2)
    e = -3 * us * us * wo / (4 * (1 - ak2 * us * us))
    f = -(1 - wo) * 3 * gc * us * us * wo / (4 * (1 - ak2 * us * us))
    dp = e / (3 * us) + us * f
    d = e + f
    b = 2 * ak / (3 - wo * 3 * gc)
    delta = np.exp(ak * taup) * (1 + b) ** 2 - np.exp(-ak * taup) * (1 - b) ** 2
    ww = wo / 4
    ss = us / (1 - ak2 * us * us)
    q1 = 2 + 3 * us + (1 - wo) * 3 * gc * us * (1 + 2 * us)
    q2 = 2 - 3 * us - (1 - wo) * 3 * gc * us * (1 - 2 * us)
    q3 = q2 * np.exp(-taup / us)
    c1 = ((ww * ss) / delta) * (q1 * np.exp(ak * taup) * (1 + b) + q3 * (1 - b))
    c2 = -((ww * ss) / delta) * (q1 * np.exp(-ak * taup) * (1 - b) + q3 * (1 + b))
    cp1 = c1 * ak / (3 - wo * 3 * gc)
    cp2 = -c2 * ak / (3 - wo * 3 * gc)
    z = d - wo * 3 * gc * uv * dp + wo * aer_phase / 4
    x = c1 - wo * 3 * gc * uv * cp1
    y = c2 - wo * 3 * gc * uv * cp2
    aa1 = uv / (1 + ak * uv)
    aa2 = uv / (1 - ak * uv)
    aa3 = us * uv / (us + uv)
    aer_ref1 = x * aa1 * (1 - np.exp(-taup / aa1))
    aer_ref2 = y * aa2 * (1 - np.exp(-taup / aa2))
    aer_ref3 = z * aa3 * (1 - np.exp(-taup / aa3))
    aer_ref = (aer_ref1 + aer_ref2 + aer_ref3) / (us * uv)
    Res_ray = (c["Resr1"] + c["Resr2"] * c["taur"] * ray_phase / (us * uv)
               + c["Resr3"] * ((c["taur"] * ray_phase / (us * uv)) ** 2))
    Res_aer = (c["Resa1"] + c["Resa2"] * (taup * m * cksi) + c["Resa3"] * ((taup * m * cksi) ** 2)) \
        + c["Resa4"] * (taup * m * cksi) ** 3
    tautot = taup + taurz
    Res_6s = (c["Rest1"] + c["Rest2"] * (tautot * m * cksi) + c["Rest3"] * ((tautot * m * cksi) ** 2)) \
        + c["Rest4"] * ((tautot * m * cksi) ** 3)
    atm_ref = ray_ref - Res_ray + aer_ref - Res_aer + Res_6s
    tdir_tts = np.exp(-tautot / us)
    tdir_ttv = np.exp(-tautot / uv)
    tdif_tts = ttetas - tdir_tts
    tdif_ttv = ttetav - tdir_ttv
    # output mapping smac.py:209-211
    return [ttetas, ttetav, tg, s, atm_ref, tdir_tts, tdif_tts, tdir_ttv, tdif_ttv]


def smac(angles, atm, sens):
    """SMAC, smac.py:14-213.

    angles (B,3) [tts,tto,psi]; atm (B,4) [aot550,uo3,uh2o,Pa]; sens = sensor_tables(...)
    returns dict of the nine AtmosphericOptics fields, each (B, nb) float64.

    All-float64.  The reference pickles hold the Sentinel-2 coefficients as float32, and under
    numpy>=2 promotion rules parts of smac.py then run in float32 (which parts depends on
    whether the caller passed Python floats or numpy scalars); that moves Ra_so by <1e-6 rel
    and R_TOA by <3e-7 rel (tests/test_oracle_golden.py), inside the 1e-6 parity budget.
    """
    angles = np.atleast_2d(np.asarray(angles, dtype=np.float64))
    atm = np.atleast_2d(np.asarray(atm, dtype=np.float64))
    B = angles.shape[0]
    coef = sens["coef"]
    nb = coef.shape[1]
    c = {n: coef[i][None, :] for i, n in enumerate(COEF_NAMES)}
    with np.errstate(all="ignore"):
        r = _smac_core(angles[:, 0:1], angles[:, 1:2], angles[:, 2:3], atm[:, 3:4],
                       atm[:, 0:1], atm[:, 1:2], atm[:, 2:3], c)
    out = [np.broadcast_to(np.asarray(v, dtype=np.float64), (B, nb)).copy() for v in r]
    return dict(zip(SMAC_OUT, out))


# --------------------------------------------------------------------------- ET radiance / SRF
def et_correction(DOY):
    """calculate_ET_radiance day-of-year factor, SPART.py:345-352."""
    b = 2 * np.pi * np.asarray(DOY, dtype=np.float64) / 365
    return 1.00011 + 0.034221 * np.cos(b) + 0.00128 * np.sin(b) + 0.000719 * np.cos(2 * b) + 0.000077 * np.sin(2 * b)


def srf_nearest_index(wl_srf):
    """get_closest_index, SPART.py:381-387: argmin_i |wl_hi[i] - v| on the 400..2400 grid.

    First minimum wins (ties at x.5 go to the lower wavelength); a NaN wavelength gives
    |.| = NaN everywhere and numpy's argmin then returns index 0.
    """
    v = np.asarray(wl_srf, dtype=np.float64)
    idx = np.ceil(v - 0.5) - 400          # nearest, ties down
    idx = np.where(np.isnan(v), 0, idx)
    return np.clip(idx, 0, NWL - 1).astype(np.int64)


def et_convolution(tables, sens):
    """Sample-independent part of calculate_spectral_convolution(wl_Ea, Ea, sensorinfo), SPART.py:358-396."""
    idx = srf_nearest_index(sens["wl_srf"])
    rad = tables["Ea"][idx]
    p = sens["p_srf"]
    return np.sum(rad * p, axis=0) / np.sum(p, axis=0)


def interp_weights(wl_smac):
    """np.interp(sensor_wavelengths, wlS, .) (SPART.py:220-223) as (i0, i1, frac)."""
    wlS = wl_solar()
    x = np.asarray(wl_smac, dtype=np.float64)
    i0 = np.clip(np.searchsorted(wlS, x, side="right") - 1, 0, NWLS - 2)
    i1 = i0 + 1
    frac = (x - wlS[i0]) / (wlS[i1] - wlS[i0])
    frac = np.clip(frac, 0.0, 1.0)      # np.interp clamps outside the grid
    return i0, i1, frac


# --------------------------------------------------------------------------- full chain
def pad_leaf(refl, tran, rho_thermal=0.01, tau_thermal=0.01):
    """set_leaf_refl_trans_assumptions, SPART.py:445-470."""
    B = refl.shape[0]
    rho = np.concatenate([refl, np.full((B, NWLT), 1.0) * np.reshape(rho_thermal, (-1, 1))], axis=1)
    tau = np.concatenate([tran, np.full((B, NWLT), 1.0) * np.reshape(tau_thermal, (-1, 1))], axis=1)
    return rho, tau


def pad_soil(refl):
    """set_soil_refl_trans_assumptions, SPART.py:427-442."""
    return np.concatenate([refl, np.repeat(refl[:, NWL - 1:NWL], NWLT, axis=1)], axis=1)


def spart_run(P, sensor, tables=None, e1="exp1", pso="quad",
              rho_thermal=0.01, tau_thermal=0.01, full=False, rdry=None, lidf=None, nlayers=NL):
    """SPART(...).run() for a fresh object per row (SPART.py:162-269).

    P : (B, 27) parameter matrix (layout in the module docstring)
    returns dict R_TOC, R_TOA, L_TOA (B, nb)  [+ intermediates when full=True]
    """
    tables = tables or load_tables()
    sens = sensor_tables(tables, sensor)
    P = np.atleast_2d(np.asarray(P, dtype=np.float64))
    leaf, soil, canopy, angles, atm, DOY = P[:, 0:9], P[:, 9:15], P[:, 15:19], P[:, 19:22], P[:, 22:26], P[:, 26]

    refl, tran, kchl = prospect_5d(leaf, tables, e1=e1)
    rwet, rdry = bsm(soil, tables, rdry=rdry)      # rdry given: SoilParametersFromFile branch (bsm.py:42-43)
    rho, tau = pad_leaf(refl, tran, rho_thermal, tau_thermal)
    rs = pad_soil(rwet)
    can = sailh(rho, tau, rs, canopy, angles, pso=pso, lidf=lidf, nlayers=nlayers)   # canopy.lidf / .nlayers (sailh.py:48, 51)
    i0, i1, fr = interp_weights(sens["wl_smac"])
    lerp = lambda y: y[:, i0] + (y[:, i1] - y[:, i0]) * fr[None, :]
    rv_so, rv_do, rv_dd, rv_sd = lerp(can["rso"]), lerp(can["rdo"]), lerp(can["rdd"]), lerp(can["rsd"])
    at = smac(angles, atm, sens)
    ta_ss, ta_sd, ta_oo, ta_do = at["Ta_ss"], at["Ta_sd"], at["Ta_oo"], at["Ta_do"]
    ra_dd, ra_so, T_g = at["Ra_dd"], at["Ra_so"], at["Tg"]
    with np.errstate(all="ignore"):
        # SPART.py:243-252
        rtoa0 = ra_so + ta_ss * rv_so * ta_oo
        rtoa1 = (ta_sd * rv_do + ta_ss * rv_sd * ra_dd * rv_do) * ta_oo / (1 - rv_dd * ra_dd)
        rtoa2 = (ta_ss * rv_sd + ta_sd * rv_dd) * ta_do / (1 - rv_dd * ra_dd)
        R_TOC = (ta_ss * rv_so + ta_sd * rv_do) / (ta_ss + ta_sd)
        R_TOA = T_g * (rtoa0 + rtoa1 + rtoa2)
        # SPART.py:180-185, 318-355, 358-396
        La = (et_correction(DOY) * np.cos(angles[:, 0] * np.pi / 180) / np.pi)[:, None] * et_convolution(tables, sens)[None, :]
        L_TOA = La * R_TOA
    out = dict(R_TOC=R_TOC, R_TOA=R_TOA, L_TOA=L_TOA)
    if full:
        out.update(leaf_refl=refl, leaf_tran=tran, kChlrel=kchl, soil_refl=rwet, soil_refl_dry=rdry,
                   rso=can["rso"], rdo=can["rdo"], rsd=can["rsd"], rdd=can["rdd"], La=La,
                   rsoil=lerp(rs), aux=can["aux"], **{"atm_" + k: v for k, v in at.items()})
    return out


