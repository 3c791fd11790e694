import sys, os, time, json, torch
sys.path.insert(0,'.')
from pyrate_amd import engine, systems, _lib
dev=torch.device('cuda',0)
recs=systems.double_gauss_records()
sysd=engine.DeviceSystem(recs,0)
o,k,e0=systems.double_gauss_bundle(10**7)
x0,k0,e0d=[engine.to_device_rays(a,dev) for a in (o,k,e0)]
out=[]
for mode in (0,1):
    bufs=sysd.alloc_outputs(x0.shape[1],mode)
    for _ in range(30): sysd.trace_into(x0,k0,bufs,e0d)
    torch.cuda.synchronize()
    res=[sysd.trace_timed(x0,k0,bufs,50,e0d) for rep in range(4)]
    out.append(("path" if mode==0 else "image", ["%.4f"%r for r in res]))
    del bufs
print("PRT_LDS_TABLE", os.environ.get("PRT_LDS_TABLE"), out)
