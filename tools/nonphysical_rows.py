import sys, os
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
sys.path[:0]=[ROOT+'/oracle',ROOT+'/tests',ROOT+'/spart-python_amd']
import numpy as np, torch, spart_oracle as o
from spart_amd import workloads, get_engine
D=workloads.default_row
rows={"LAI=0":D(LAI=0.0),"q=0":D(q=0.0),"tts=90":D(tts=90.0),"LAI=-1":D(LAI=-1.0),"Pa=0":D(Pa=0.0),"default":D(),"N=0.5":D(N=0.5),"N=0.9":D(N=0.9),"LAI=1e-9":D(LAI=1e-9)}
P=np.concatenate(list(rows.values()))
t=o.load_tables()
with np.errstate(all="ignore"):
    ref=o.spart_run(P,"Sentinel2A-MSI",t,pso="gl")
out=get_engine("Sentinel2A-MSI",0).run(torch.as_tensor(P.T.copy(),device="cuda:0"),"float64")
for i,n in enumerate(rows):
    print(n, {k: float(np.max(np.abs(out[k][i].cpu().numpy()-ref[k][i])/np.maximum(np.abs(ref[k][i]),1e-6))) for k in ("R_TOC","R_TOA","L_TOA")})
