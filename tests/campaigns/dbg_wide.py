import sys, numpy as np, torch
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import _golden
from oracle import seqtrace_np as oracle
from pyrate_amd import engine
name = sys.argv[1] if len(sys.argv)>1 else "double_gauss_wide"
case=_golden.load_case(name)
dev=torch.device("cuda",0)
sysd=engine.DeviceSystem(case.table,0)
e=np.asarray(case.E0)
res=sysd.trace(engine.to_device_rays(case.x0,dev),engine.to_device_rays(case.k0,dev),engine.to_device_rays(e.real,dev), engine.to_device_rays(e.imag,dev) if np.iscomplexobj(e) else None)
torch.cuda.synchronize()
dense=_golden.dense_from_engine(res)
out=oracle.trace(case.table,case.x0,case.k0,case.E0)
np.set_printoptions(precision=17, linewidth=200)
for s in range(case.n_surfaces):
    v=out[s]["valid"]; vd=dense[s]["valid"].astype(bool)
    ex=np.abs(dense[s]["x_hit"]-out[s]["x_hit"]).max(0)
    ek=np.abs(dense[s]["k_out"]-np.real(out[s]["k_out"])).max(0)
    exv=np.where(v,ex,0); 
    j=int(np.nanargmax(np.nan_to_num(exv)))
    print(s, "valid eq",np.array_equal(v,vd), "maxex(valid)",np.nanmax(exv), "ray",j, "ek", np.nanmax(np.where(out[s]["valid_out"][:len(ek)] if len(ek)==len(out[s]["valid_out"]) else True,ek,0)))
    if np.nanmax(exv)>1e-9:
        print("  oracle x",out[s]["x_hit"][:,j]," hip x",dense[s]["x_hit"][:,j])
        if s>0: print("  prev oracle x",out[s-1]["x_hit"][:,j], "k", out[s-1]["k_out"][:,j], " hip prev x",dense[s-1]["x_hit"][:,j], "k",dense[s-1]["k_out"][:,j], "valid_out prev", out[s-1]["valid_out"][j], dense[s-1]["valid_out"][j])
        break
print("---- golden compare")
b=case.bundles
pos=np.array(b[1]["id"])
for s in range(case.n_surfaces):
    B=b[s+1]; d=dense[s]; Bn=b[s+2]
    xr=B["x"][-1]; vr=B["valid"][-1].astype(bool)
    xd=d["x_hit"][:,pos]; vd=d["valid"][pos].astype(bool)
    m=vr&vd
    err=np.abs(xd[:,m]-xr[:,m]).max(0) if m.any() else np.zeros(0)
    print(s,"n_ref",len(pos),"valid eq",np.array_equal(vr,vd),"max err",err.max() if len(err) else 0)
    if len(err) and err.max()>1e-9:
        j=np.where(m)[0][int(np.argmax(err))]
        print("   ref slot",j,"dense slot",pos[j],"ref x",xr[:,j],"hip x",xd[:,j],"oracle x",out[s]["x_hit"][:,pos[j]], "oracle valid", out[s]["valid"][pos[j]])
    sub=_golden._subsequence_positions(B["x"][-1],B["id"],Bn["x"][0],Bn["id"])
    pos=pos[sub]
