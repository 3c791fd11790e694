"""where does a merit evaluation of the asphere optimisation demo spend its host time?"""
import cProfile, pstats, sys, io
sys.path.insert(0, '.')
from demos import demo_optimize_asphere
demo_optimize_asphere.main(maxiter=30)          # warm everything
pr = cProfile.Profile()
pr.enable()
demo_optimize_asphere.main(maxiter=200)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
