import sys, time
sys.path.insert(0, "spart-python_amd")
import SPART, torch, numpy as np
leafbio = SPART.LeafBiology(40, 0.01, 0.02, 0, 10, 10, 1.5)
soilpar = SPART.SoilParameters(0.5, 0, 100, 15, 25, 0.015)
canopy  = SPART.CanopyStructure(3, -0.35, -0.15, 0.05)
angles  = SPART.Angles(40, 0, 0)
atm     = SPART.AtmosphericProperties(0.3246, 0.3480, 1.4116, 1013.25)
sp = SPART.SPART(soilpar, leafbio, canopy, atm, angles, "Sentinel2A-MSI", 100)
sp.run()
t=time.perf_counter()
for i in range(200):
    sp.leafbio = SPART.LeafBiology(40+i*0.01, 0.01, 0.02, 0, 10, 10, 1.5)
    df = sp.run()
dt=(time.perf_counter()-t)/200
print("scalar SPART.run(): %.3f ms per call" % (dt*1e3))
import cProfile, pstats
pr=cProfile.Profile(); pr.enable()
for i in range(100): sp.run()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
