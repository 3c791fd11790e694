"""host-time profile of the --fast merit evaluation (image_moments) of the asphere optimisation demo"""
import cProfile, pstats, sys, io, time
sys.path.insert(0, '.')
from demos import demo_optimize_asphere
demo_optimize_asphere.main(maxiter=60, fast=True)
pr = cProfile.Profile()
pr.enable()
demo_optimize_asphere.main(maxiter=400, fast=True)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:5000])
