import sys, cProfile, pstats, torch
sys.path.insert(0,'.')
from pyrate_amd import systems
from pyrate_amd.builders import build_rotationally_symmetric_optical_system
from pyrate_amd.raytracer.ray import RayBundle
(s, seq) = build_rotationally_symmetric_optical_system(systems.double_gauss_tuples())
(o, k, e0) = systems.double_gauss_bundle(100)
ib = RayBundle(o, k, e0, wave=systems.DLINE)
for _ in range(20): s.seqtrace(ib, seq)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300):
    rp = s.seqtrace(ib, seq)
    x = rp[0].raybundles[-1].x
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
