"""march time for x_hit from allocation i and k_out from allocation j (8 x 8), same process"""
import sys, torch
sys.path.insert(0, '.')
from pyrate_amd import engine, systems, _lib
dev = torch.device("cuda", 0)
sysd = engine.DeviceSystem(systems.double_gauss_records(), 0)
(x0, k0, e0d, _) = systems.double_gauss_bundle_device(10000000, dev)
n = x0.shape[1]
K = 8
sets = [sysd.alloc_outputs(n, _lib.MODE_PATH, packed_flags=True) for _ in range(K)]
sysd.trace_timed(x0, k0, sets[0], 40, e0d)
def march(b):
    sysd.trace_timed(x0, k0, b, 2, e0d)
    return sysd.trace_timed(x0, k0, b, 6, e0d)
print("rows: x_hit from set i; columns: k_out from set j (flags from set 0)")
for i in range(K):
    row = []
    for j in range(K):
        b = dict(sets[0]); b["x_hit"] = sets[i]["x_hit"]; b["k_out"] = sets[j]["k_out"]
        row.append("%.3f" % march(b))
    print("x%d: " % i + " ".join(row), flush=True)
# swap roles: write k into an 'x' allocation and x into a 'k' allocation
print("x_hit := k_out buffer of set j, k_out := x_hit buffer of set j")
print(" ".join("%.3f" % march(dict(sets[0], x_hit=sets[j]["k_out"], k_out=sets[j]["x_hit"])) for j in range(K)))
print("plain sets")
print(" ".join("%.3f" % march(sets[j]) for j in range(K)))
