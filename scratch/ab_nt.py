# A/B of non-temporal loads/stores in k_trace_iso.  Needs an experimental build that exports
# prt_set_experiment(flags) (bit0 NT stores, bit1 NT loads) -- the switch is not part of the shipped
# kernel; results are quoted in DESIGN.md section 5.
import sys, ctypes, torch
sys.path.insert(0,'.')
from pyrate_amd import engine, systems, _lib
lib=_lib.load()
lib_raw=ctypes.CDLL(_lib.LIB_PATH)
dev=torch.device('cuda',0)
sysd=engine.DeviceSystem(systems.double_gauss_records(),0)
x0,k0,e0,_=systems.double_gauss_bundle_device(10**7,dev)
bufs=sysd.alloc_outputs(x0.shape[1],0)
for _ in range(30): sysd.trace_into(x0,k0,bufs,e0)
torch.cuda.synchronize()
names={0:"base",1:"NT store",2:"NT load",3:"NT both"}
res={k:[] for k in names}
for rep in range(6):
    for f in (0,1,2,3):
        lib_raw.prt_set_experiment(f)
        res[f].append(sysd.trace_timed(x0,k0,bufs,40,e0))
for f in names: print("%-9s"%names[f], " ".join("%.4f"%v for v in res[f]))
