import sys, time, torch
sys.path.insert(0, '.')
from pyrate_amd import systems
from pyrate_amd.builders import build_rotationally_symmetric_optical_system
from pyrate_amd.raytracer.ray import RayBundle
(s, seq) = build_rotationally_symmetric_optical_system(systems.double_gauss_tuples())
for nr in (100, 100, 10000, 100):
    (o, k, e0) = systems.double_gauss_bundle(nr)
    ib = RayBundle(o, k, e0, wave=systems.DLINE)
    s.seqtrace(ib, seq); torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); rp = s.seqtrace(ib, seq); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(nr, o.shape[1], " ".join("%.2f" % t for t in ts))
