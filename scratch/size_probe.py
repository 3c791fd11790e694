"""per-ray cost of the path-mode march vs bundle size, one process, interleaved repeats"""
import sys, torch
sys.path.insert(0, '.')
from pyrate_amd import engine, systems, _lib
dev = torch.device("cuda", 0)
sysd = engine.DeviceSystem(systems.double_gauss_records(), 0)
sizes = (2500000, 5000000, 10000000, 12500000, 20000000, 40000000)
data = {}
for nr in sizes:
    (x0, k0, e0d, _) = systems.double_gauss_bundle_device(nr, dev)
    n = x0.shape[1]
    data[nr] = (x0, k0, e0d, sysd.alloc_outputs(n, _lib.MODE_PATH, packed_flags=True))
(x0, k0, e0d, b) = data[10000000]
sysd.trace_timed(x0, k0, b, 40, e0d)
for rep in range(4):
    line = []
    for nr in sizes:
        (x0, k0, e0d, b) = data[nr]
        n = x0.shape[1]
        sysd.trace_timed(x0, k0, b, 3, e0d)
        ms = sysd.trace_timed(x0, k0, b, max(4, int(2e8 / n)), e0d)
        line.append("%.1e: %.4f ms %.5f ns/ray" % (n, ms, ms * 1e6 / n))
    print(" | ".join(line), flush=True)
