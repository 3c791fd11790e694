import sys, time, torch
sys.path.insert(0,'.')
from pyrate_amd import engine, systems
dev=torch.device('cuda',0)
recs=systems.double_gauss_records()
sysd=engine.DeviceSystem(recs,0)
x0,k0,e0,_=systems.double_gauss_bundle_device(10**7,dev)
x0=x0.contiguous(); k0=k0.contiguous(); e0=e0.contiguous()
n=x0.shape[1]
def t(f,reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/reps
xh,v=sysd.propagate(0,x0,k0,e_re=e0)
ms=t(lambda: sysd.propagate(0,x0,k0,e_re=e0)); print("propagate (E given): %.3f ms  %.2f TB/s"%(ms, n*(72+25)/ms/1e9))
ms=t(lambda: sysd.propagate(1,xh,k0,default_e=False,valid_in=v)); print("propagate (d=k/|k|): %.3f ms  %.2f TB/s"%(ms, n*(49+25)/ms/1e9))
ms=t(lambda: sysd.interact(0,xh,k0,valid_in=v)); print("interact iso: %.3f ms  %.2f TB/s (incl. dir_out)"%(ms, n*(49+49)/ms/1e9))
k2,d,vo,_,_=sysd.interact(0,xh,k0,valid_in=v)
ms=t(lambda: engine.compact(vo,[xh,k2],None),5); print("compact 6 rows: %.3f ms  %.2f TB/s"%(ms, n*(1+96)/ms/1e9))
cnt=t(lambda: engine.bundle_moments(xh, vo),5); print("moments: %.3f ms  %.2f TB/s"%(cnt, n*25/cnt/1e9))
