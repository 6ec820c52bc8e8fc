"""The reference's quick start (one parameter set -> the reference's DataFrame), then the same call with arrays:
a whole look-up table in one launch.  Needs an MI355X (there is no CPU path).

    python examples/example.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "spart-python_amd"))
import SPART  # noqa: E402

leafbio = SPART.LeafBiology(Cab=40, Cca=10, Cw=0.02, Cdm=0.01, Cs=0, Cant=10, N=1.5)
soilpar = SPART.SoilParameters(B=0.5, lat=0, lon=100, SMp=20, SMC=25, film=0.015)
canopy = SPART.CanopyStructure(LAI=3, LIDFa=-0.35, LIDFb=-0.15, q=0.05)
angles = SPART.Angles(sol_angle=40, obs_angle=0, rel_angle=0)
atm = SPART.AtmosphericProperties(aot550=0.325, uo3=0.35, uh2o=1.41, Pa=1013.25)

# scalars in -> pandas DataFrame with Band, L_TOA, R_TOA, R_TOC, indexed by the band centres
print(SPART.SPART(soilpar, leafbio, canopy, atm, angles, sensor="Sentinel2A-MSI", DOY=100).run())

# any field may be an array of length B: 100 000 canopies, LAI and chlorophyll varying
B = 100_000
rng = np.random.default_rng(0)
leafbio = SPART.LeafBiology(Cab=rng.uniform(10, 80, B), Cca=10, Cw=0.02, Cdm=0.01, Cs=0, Cant=10, N=1.5)
canopy = SPART.CanopyStructure(LAI=rng.uniform(0.1, 7, B), LIDFa=-0.35, LIDFb=-0.15, q=0.05)
res = SPART.SPART(soilpar, leafbio, canopy, atm, angles, sensor="Sentinel2A-MSI", DOY=100, dtype="float32").run()
print(type(res).__name__, res["R_TOC"].shape, res["R_TOA"].mean(axis=0))

# the model's tables are the object's public attributes (SPART.py:93-95) and are read on EVERY run() here: a 5 % stronger
# chlorophyll absorption and a Sentinel-2 band moved by 3.5 nm, no new object needed.  (Upstream reads them only while its change
# trackers are set, SPART.py:178-209: after a first run() this very edit would return the first answer again until a setter
# such as `sp.leafbio = sp.leafbio` flips a tracker -- this package is stateless by design, see SPART.run.__doc__.)
sp = SPART.SPART(soilpar, SPART.LeafBiology(40, 0.01, 0.02, 0, 10, 10, 1.5), SPART.CanopyStructure(3, -0.35, -0.15, 0.05), atm, angles,
                 sensor="Sentinel2A-MSI", DOY=100)
before = sp.run()
sp.optipar["Kab"] *= 1.05
sp.sensorinfo["wl_smac"] = sp.sensorinfo["wl_smac"].astype(float) + 3.5
after = sp.run()
print("R_TOC at the red band: %.5f -> %.5f (centre %g -> %g nm)" % (before["R_TOC"].iloc[3], after["R_TOC"].iloc[3], before.index[3], after.index[3]))

# the canopy state SAILH reads from the object (sailh.py:48, 51): a measured leaf-angle distribution instead of (LIDFa, LIDFb),
# and another number of canopy layers
canopy = SPART.CanopyStructure(LAI=3, LIDFa=-0.35, LIDFb=-0.15, q=0.05)
canopy.lidf = np.array([0.22, 0.19, 0.16, 0.13, 0.10, 0.07, 0.05, 0.03, 0.02, 0.01, 0.01, 0.005, 0.005])[:, None]   # 13 classes, sums to 1
canopy.nlayers = 30
sp = SPART.SPART(soilpar, SPART.LeafBiology(40, 0.01, 0.02, 0, 10, 10, 1.5), canopy, atm, angles, sensor="Sentinel2A-MSI", DOY=100)
print("planophile table, 30 layers: R_TOC(NIR) = %.5f (spherical-ish default: %.5f)" % (sp.run()["R_TOC"].iloc[8], before["R_TOC"].iloc[8]))
