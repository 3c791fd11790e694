"""packed mask byte vs two mask arrays on the SAME x_hit / k_out arrays, one process"""
import sys, torch
sys.path.insert(0, '.')
from pyrate_amd import engine, systems, _lib
dev = torch.device("cuda", 0)
sysd = engine.DeviceSystem(systems.double_gauss_records(), 0)
(x0, k0, e0d, _) = systems.double_gauss_bundle_device(10000000, dev)
n = x0.shape[1]
(packed, rep) = sysd.alloc_outputs_tuned(x0, k0, e0d, packed_flags=True)
print("placement: first %.4f chosen %.4f" % (rep["first_pair_ms"], rep["best_pair_ms"]))
two = dict(packed, packed_flags=False, valid_out=torch.empty_like(packed["valid"]))
sysd.trace_timed(x0, k0, packed, 40, e0d)
for r in range(4):
    a = (sysd.trace_timed(x0, k0, packed, 2, e0d), sysd.trace_timed(x0, k0, packed, 20, e0d))[1]
    b = (sysd.trace_timed(x0, k0, two, 2, e0d), sysd.trace_timed(x0, k0, two, 20, e0d))[1]
    print("packed %.4f ms  two arrays %.4f ms  (%.2f %%)" % (a, b, (b / a - 1) * 100), flush=True)
