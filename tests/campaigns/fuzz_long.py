"""one-off long fuzz: the parametrised fuzz tests with many more seeds"""
import sys, traceback
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import test_gpu_fuzz as tf
dev = torch.device("cuda", 0)
bad = []
for seed in range(24, 424):
    try:
        tf.test_random_systems_match_oracle(dev, seed)
    except Exception as exc:
        bad.append(("systems", seed, repr(exc)[:300]))
for seed in range(12, 132):
    try:
        tf.test_random_crystals_match_oracle(dev, seed)
    except Exception as exc:
        bad.append(("crystals", seed, repr(exc)[:300]))
print("failures:", len(bad))
for b in bad[:20]:
    print(b)
