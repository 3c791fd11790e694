"""drop-in OpticalSystem.seqtrace at 1e7 rays vs the bench kernel (round 2 verdict item 1)"""
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
print(json.dumps({"dropin_seqtrace_call_ms": ts, "rays": o.shape[1],
                  "arena": placed.PlacedArena.for_device(0).stats()}))
