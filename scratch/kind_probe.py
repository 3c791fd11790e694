"""does an array's 'kind' (fast / slow partner in the march) show up in a translation-heavy access pattern?
strided single-line writes (one 8-B store per 4 KiB / 64 KiB / 2 MiB) over each candidate array"""
import sys, torch
sys.path.insert(0, '.')
from pyrate_amd import engine, systems, _lib
dev = torch.device("cuda", 0)
sysd = engine.DeviceSystem(systems.double_gauss_records(), 0)
(x0, k0, e0d, _) = systems.double_gauss_bundle_device(10000000, dev)
n = x0.shape[1]
first = sysd.alloc_outputs(n, _lib.MODE_PATH, packed_flags=True)
words = first["x_hit"].numel()
pool = [first["x_hit"], first["k_out"]]
sp = []
for q in range(8):
    sp.append(torch.empty(int(20e9), dtype=torch.uint8, device=dev))
    pool.append(torch.empty(words, dtype=torch.float64, device=dev))
sysd.trace_timed(x0, k0, first, 40, e0d)
def march(i, j):
    b = dict(first, x_hit=pool[i], k_out=pool[j])
    sysd.trace_timed(x0, k0, b, 1, e0d)
    return sysd.trace_timed(x0, k0, b, 4, e0d)
def strided(t, stride_words, iters=20):
    v = t[::stride_words]
    v.fill_(1.0); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        v.fill_(2.0)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3      # microseconds
M = len(pool)
print("march with x = array i, k = array (i+1) %% M, and with k = array 0 / 1:")
for i in range(M):
    row = "a%-2d pairs: next %.3f  with a0 %s  with a1 %s" % (i, march(i, (i + 1) % M),
          ("%.3f" % march(i, 0)) if i != 0 else "  -  ", ("%.3f" % march(i, 1)) if i != 1 else "  -  ")
    row += " | strided writes us: 4KiB %.1f  64KiB %.1f  2MiB %.1f" % (strided(pool[i], 512), strided(pool[i], 8192), strided(pool[i], 262144))
    print(row, flush=True)
