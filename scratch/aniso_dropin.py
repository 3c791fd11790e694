import sys, time, math, torch, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import systems_zoo as zoo
from pyrate_amd import systems
api=zoo.mirror_api()
c=systems.CALCITE_TILTED
eps1=systems.uniaxial_eps(c["n_o"],c["n_e"],c["axis"]); eps2=systems.uniaxial_eps(1.6727,1.60,(math.sin(0.2),0.0,math.cos(0.2)))
(s,seq)=zoo.aniso_doublet(api,eps1,eps2)
for n in (10**4,10**6):
    (o,k)=systems.collimated_bundle(n,11.43,-5.0)
    e0=np.cross(k,np.array([1.,0,0]),axisa=0,axisb=0).T.copy()
    ib=api.RayBundle(o,k,e0,wave=0.5876e-3)
    s.seqtrace(ib,seq); torch.cuda.synchronize()
    t=time.perf_counter()
    for _ in range(5): rp=s.seqtrace(ib,seq)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/5
    t=time.perf_counter(); x=rp[0].raybundles[-1].x; dt2=time.perf_counter()-t
    print(o.shape[1], "seqtrace %.2f ms"%(dt*1e3), "touch image %.2f ms"%(dt2*1e3), x.shape)
