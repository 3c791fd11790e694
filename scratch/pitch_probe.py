"""Does the path-mode march depend on (a) which allocation the outputs land in, (b) the row pitch's
factor of two?  Same process, fresh buffers per measurement."""
import sys, torch
sys.path.insert(0, '.')
from pyrate_amd import engine, systems, _lib
dev = torch.device("cuda", 0)
sysd = engine.DeviceSystem(systems.double_gauss_records(), 0)

def v2(x):
    k = 0
    while x % 2 == 0:
        x //= 2; k += 1
    return k

for nrays in (10000000, 12500000):
    (x0, k0, e0d, _) = systems.double_gauss_bundle_device(nrays, dev)
    n = x0.shape[1]
    base = (n + 511) // 512
    cands = [base, base + 1, base + 2, base + 3, (base + 15) // 16 * 16, (base + 15) // 16 * 16 + 1,
             (base + 255) // 256 * 256, (base + 255) // 256 * 256 + 1, base]
    warm = sysd.alloc_outputs(n, _lib.MODE_PATH, packed_flags=True)
    sysd.trace_timed(x0, k0, warm, 30, e0d)
    del warm
    for rep in range(2):
        keep = []
        for q in cands:
            bufs = sysd.alloc_outputs(n, _lib.MODE_PATH, packed_flags=True, pitch=q * 512)
            sysd.trace_timed(x0, k0, bufs, 5, e0d)
            ms = sysd.trace_timed(x0, k0, bufs, 20, e0d)
            print("n %d pitch %d*512 (2^%d) : %.4f ms  %.5f ns/ray  x_hit@%x" % (n, q, v2(q), ms, ms * 1e6 / n, bufs["x_hit"].data_ptr()), flush=True)
            keep.append(bufs)          # keep them alive so the next one gets a different allocation
        del keep
        torch.cuda.empty_cache()
