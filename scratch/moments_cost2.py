"""plain march vs march with fused spot moments on the SAME arrays (one process)"""
import sys, torch
sys.path.insert(0, '.')
from pyrate_amd import engine, systems, _lib
dev = torch.device("cuda", 0)
sysd = engine.DeviceSystem(systems.double_gauss_records(), 0)
(x0, k0, e0d, _) = systems.double_gauss_bundle_device(12500000, dev)
n = x0.shape[1]
(bufs, rep) = sysd.alloc_outputs_tuned(x0, k0, e0d, packed_flags=True)
ws = engine.MomentsWorkspace(dev, n_results=1, n_rays=n)
def timed(fn, iters=30):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    for _ in range(3): fn()
    torch.cuda.synchronize(); a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
plain = lambda: sysd.trace_into(x0, k0, bufs, e0d)
fused = lambda: sysd.trace_moments_into(x0, k0, bufs, ws, slot=0, e0_re=e0d)
timed(plain, 40)
for r in range(3):
    print("plain %.4f ms   with fused moments %.4f ms" % (timed(plain), timed(fused)), flush=True)
