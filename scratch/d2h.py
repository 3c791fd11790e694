"""drop-in seqtrace: cost of touching NumPy views of the results (device compaction + D2H)"""
import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from pyrate_amd import systems, builders
from pyrate_amd.raytracer.ray import RayBundle
(s, seq) = builders.build_rotationally_symmetric_optical_system(systems.double_gauss_tuples())
for nr in (10**6, 10**7):
    (o, k, e0) = systems.double_gauss_bundle(nr)
    ib = RayBundle(o, k, e0, wave=0.5876e-3)
    for rep in range(3):
        rp = s.seqtrace(ib, seq)[0]
        torch.cuda.synchronize()
        t = time.perf_counter(); x = rp.raybundles[-1].x; t1 = time.perf_counter() - t
        t = time.perf_counter(); kk = rp.raybundles[-1].k; t2 = time.perf_counter() - t
        t = time.perf_counter(); x5 = rp.raybundles[5].x; t3 = time.perf_counter() - t
        print(nr, "image x %.2f ms  image k %.2f ms  bundle 5 x (2 points) %.2f ms" % (t1 * 1e3, t2 * 1e3, t3 * 1e3), x.shape, x5.shape)
