"""is 'slow' a property of the individual buffer?  fill_ bandwidth per buffer, march time per set,
and crossed sets (x_hit of one, k_out of another)"""
import sys, torch
sys.path.insert(0, '.')
from pyrate_amd import engine, systems, _lib
dev = torch.device("cuda", 0)
sysd = engine.DeviceSystem(systems.double_gauss_records(), 0)
(x0, k0, e0d, _) = systems.double_gauss_bundle_device(10000000, dev)
n = x0.shape[1]
sets = [sysd.alloc_outputs(n, _lib.MODE_PATH, packed_flags=True) for _ in range(8)]

def t_fill(t, iters=10):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    t.fill_(1.0); torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        t.fill_(1.0)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

def march(b):
    sysd.trace_timed(x0, k0, b, 3, e0d)
    return sysd.trace_timed(x0, k0, b, 15, e0d)

sysd.trace_timed(x0, k0, sets[0], 40, e0d)
ms = [march(b) for b in sets]
print("march per set   : " + " ".join("%.3f" % m for m in ms))
gb = sets[0]["x_hit"].numel() * 8 / 1e9
print("fill x_hit TB/s : " + " ".join("%.2f" % (gb / t_fill(b["x_hit"])) for b in sets))
print("fill k_out TB/s : " + " ".join("%.2f" % (gb / t_fill(b["k_out"])) for b in sets))
ms2 = [march(b) for b in sets]
print("march again     : " + " ".join("%.3f" % m for m in ms2))
order = sorted(range(len(sets)), key=lambda i: ms[i])
(fast, slow) = (order[0], order[-1])
print("fastest set %d, slowest set %d" % (fast, slow))
for (tag, xi, ki, vi) in (("x slow, k fast, v fast", slow, fast, fast), ("x fast, k slow, v fast", fast, slow, fast),
                          ("x fast, k fast, v slow", fast, fast, slow), ("x slow, k slow, v fast", slow, slow, fast)):
    b = dict(sets[fast])
    b["x_hit"] = sets[xi]["x_hit"]; b["k_out"] = sets[ki]["k_out"]; b["valid"] = sets[vi]["valid"]
    print("%-24s: %.3f" % (tag, march(b)))
# image mode and a read-only pass for comparison
