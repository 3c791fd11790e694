"""one-off: crystals with NON-symmetric real epsilon tensors (the reference accepts any 3x3) vs oracle"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import test_gpu_fuzz as tf
from pyrate_amd import systems
dev = torch.device("cuda", 0)
bad = []; tot = 0
for seed in range(200):
    rng = np.random.RandomState(99000 + seed)
    def eps():
        return np.eye(3) * rng.uniform(2.0, 3.0) + rng.uniform(-0.25, 0.25, (3, 3))
    recs = systems.aniso_doublet_records(eps(), eps())
    (x0, k0, e0) = tf.wide_bundle(rng, 150, 8.0, 0.2)
    try:
        tot += tf.compare_with_oracle(recs, x0, k0, e0, dev, tight=True, tol=1e-8)
    except AssertionError as exc:
        bad.append((seed, str(exc)[:200]))
print("compared:", tot, " failures:", len(bad))
for b in bad[:20]: print(b)
