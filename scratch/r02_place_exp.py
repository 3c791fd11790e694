"""placement experiments on the double Gauss march (1e7 x 12): where do inputs / masks want to live?"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyrate_amd import engine, systems, placed, _lib
dev = torch.device("cuda", 0)
sysd = engine.DeviceSystem(systems.double_gauss_records(), 0)
(x0, k0, e0d, n) = systems.double_gauss_bundle_device(10000000, dev)
alg = n * (72 + 49 * 12)
arena = placed.PlacedArena.for_device(0)
def frac(ms): return round(alg / (ms * 1e-3) / 8e12, 4)
def timed(x, k, e, bufs):
    sysd.trace_timed(x, k, bufs, 10, e)
    return min(sysd.trace_timed(x, k, bufs, 20, e) for _ in range(3))
out = {}
warm = sysd.alloc_outputs(n, packed_flags=True, placement="torch")
for _ in range(30): sysd.trace_into(x0, k0, warm, e0d)
out["torch_outputs"] = frac(timed(x0, k0, e0d, warm)); del warm
pitch = x0.stride(0)
b = sysd.alloc_outputs(n, packed_flags=True, placement="arena", extra_bytes=[9 * pitch * 8])
out["arena_outputs_torch_inputs"] = frac(timed(x0, k0, e0d, b))
ext = b["extra"][0][:9 * pitch * 8].view(torch.float64).view(9, pitch)
xi = ext[0:3, :n]; ki = ext[3:6, :n]; ei = ext[6:9, :n]
xi.copy_(x0); ki.copy_(k0); ei.copy_(e0d)
out["arena_outputs_arena_inputs"] = frac(timed(xi, ki, ei, b))
out["kinds_x_k_inputs"] = b["placement"]["kinds"]
# masks with k_out instead of with x_hit
b2 = dict(b)
nk = b["k_out"].numel() * 8
out["note"] = "masks moved behind k_out needs a bigger k part; skipped" 
out["arena"] = arena.stats()
print(json.dumps(out))
