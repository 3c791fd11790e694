"""drop-in OpticalSystem.seqtrace at 1e7 rays vs the bench kernel (round 2 verdict item 1): one call at a
time (host work + kernel, synchronised) and back to back (the host work of call i+1 overlaps the kernel of
call i: what an optimiser loop or a wavelength sweep sees)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyrate_amd import engine, systems, placed
from pyrate_amd.builders import build_rotationally_symmetric_optical_system
from pyrate_amd.raytracer.ray import RayBundle
(s, seq) = build_rotationally_symmetric_optical_system(systems.double_gauss_tuples())
(o, k, e0) = systems.double_gauss_bundle(10000000)
ib = RayBundle(o, k, e0, wave=systems.DLINE)
for _ in range(5):
    rp = s.seqtrace(ib, seq); del rp
torch.cuda.synchronize()
ts = []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rp = s.seqtrace(ib, seq)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    del rp
for _ in range(6):                      # two result sets are alive in this loop: let the arena build the second
    rp = s.seqtrace(ib, seq)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    rp = s.seqtrace(ib, seq)
torch.cuda.synchronize()
back_to_back = (time.perf_counter() - t0) * 1e3 / 20
# the kernel alone into the same kind of arrays
sysd = engine.DeviceSystem(systems.double_gauss_records(), 0)
(x0, k0, e0d) = [engine.to_device_rays(a, torch.device("cuda", 0)) for a in (o, k, e0)]
bufs = sysd.alloc_outputs(o.shape[1], packed_flags=True)
sysd.trace_timed(x0, k0, bufs, 10, e0d)
kernel_ms = sysd.trace_timed(x0, k0, bufs, 30, e0d)
print(json.dumps({"dropin_seqtrace_call_ms_synchronised": [round(t, 4) for t in ts],
                  "dropin_seqtrace_ms_back_to_back": back_to_back, "kernel_ms": kernel_ms, "rays": o.shape[1],
                  "arena": placed.PlacedArena.for_device(0).stats()}))
