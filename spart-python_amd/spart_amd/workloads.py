"""Synthetic parameter batches for the benchmark configurations (BASELINE.json configs).

A batch is a (B, 27) float64 matrix, one row per spectrum, columns

    leaf   0..8   Cab, Cdm, Cw, Cs, Cca, Cant, N, PROT, CBC
    soil   9..14  B, lat, lon, SMp, SMC, film
    canopy 15..18 LAI, LIDFa, LIDFb, q
    angles 19..21 tts, tto, psi
    atm    22..25 aot550, uo3, uh2o, Pa
    DOY    26

drawn from a Latin hypercube (scipy.stats.qmc.LatinHypercube, seed 20240613) over the varying
columns only and scaled to the uniform ranges below; fixed columns are constants.
"""
import numpy as np

PARAM_NAMES = [
    "Cab", "Cdm", "Cw", "Cs", "Cca", "Cant", "N", "PROT", "CBC",
    "B", "lat", "lon", "SMp", "SMC", "film",
    "LAI", "LIDFa", "LIDFb", "q",
    "tts", "tto", "psi",
    "aot550", "uo3", "uh2o", "Pa",
    "DOY",
]
NPARAM = len(PARAM_NAMES)
LHS_SEED = 20240613

RANGES = dict(
    Cab=(10, 80), Cdm=(0.002, 0.02), Cw=(0.005, 0.05), Cs=(0, 0.5), Cca=(2, 20), Cant=(0, 10), N=(1, 3),
    PROT=(0, 0.003), CBC=(0, 0.01),
    B=(0.3, 0.9), lat=(-30, 30), lon=(80, 120), SMp=(5, 55),
    LAI=(0.1, 7), LIDFa=(-0.5, 0.5), LIDFb=(-0.3, 0.3), q=(0.01, 0.2),
    tts=(0, 60), tto=(0, 30), psi=(0, 180),
    aot550=(0.05, 0.5), uo3=(0.25, 0.45), uh2o=(0.5, 4), Pa=(950, 1030),
)
FIXED = dict(PROT=0.0, CBC=0.0, SMC=25.0, film=0.015, DOY=100.0)

LEAF5D = ["Cab", "Cdm", "Cw", "Cs", "Cca", "Cant", "N"]
LEAFPRO = ["Cab", "Cw", "Cs", "Cca", "Cant", "N", "PROT", "CBC"]
REST = ["B", "lat", "lon", "SMp", "LAI", "LIDFa", "LIDFb", "q", "tts", "tto", "psi", "aot550", "uo3", "uh2o", "Pa"]

# defaults of the reference's test fixtures (tests/conftest.py:90-112) for the non-varying groups
DEFAULTS = dict(
    Cab=40, Cdm=0.01, Cw=0.02, Cs=0, Cca=10, Cant=10, N=1.5, PROT=0.0, CBC=0.0,
    B=0.5, lat=0, lon=100, SMp=20, SMC=25, film=0.015,
    LAI=3, LIDFa=-0.35, LIDFb=-0.15, q=0.05,
    tts=40, tto=0, psi=0, aot550=0.325, uo3=0.35, uh2o=1.41, Pa=1013.25, DOY=100,
)


def varying_columns(kind):
    if kind == "leaf":          # config 2: PROSPECT-5D leaf only, d = 7
        return list(LEAF5D)
    if kind == "full":          # configs 3/4: d = 22
        return LEAF5D + REST
    if kind == "pro":           # config 5: Cdm = 0, PROT/CBC vary, d = 23
        return LEAFPRO + REST
    raise ValueError(kind)


def lhs_params(n, kind="full", seed=LHS_SEED):
    """(n, 27) float64 parameter matrix for BASELINE config ``kind`` ('leaf' | 'full' | 'pro')."""
    from scipy.stats import qmc

    cols = varying_columns(kind)
    u = qmc.LatinHypercube(d=len(cols), seed=seed).random(n)
    lo = np.array([RANGES[c][0] for c in cols], dtype=np.float64)
    hi = np.array([RANGES[c][1] for c in cols], dtype=np.float64)
    x = qmc.scale(u, lo, hi)
    P = np.empty((n, NPARAM), dtype=np.float64)
    for j, name in enumerate(PARAM_NAMES):
        if name in cols:
            P[:, j] = x[:, cols.index(name)]
        elif kind == "pro" and name == "Cdm":
            P[:, j] = 0.0
        elif name in FIXED:
            P[:, j] = FIXED[name]
        else:
            P[:, j] = DEFAULTS[name]
    return P


def default_row(**over):
    """One row with the reference's fixture defaults, optionally overridden by name."""
    d = dict(DEFAULTS)
    d.update(over)
    return np.array([[float(d[n]) for n in PARAM_NAMES]], dtype=np.float64)
