"""march time for x_hit / k_out arrays spread over the whole HBM (spacers between the arrays)"""
import sys, torch
sys.path.insert(0, '.')
from pyrate_amd import engine, systems, _lib
dev = torch.device("cuda", 0)
sysd = engine.DeviceSystem(systems.double_gauss_records(), 0)
(x0, k0, e0d, _) = systems.double_gauss_bundle_device(10000000, dev)
n = x0.shape[1]
first = sysd.alloc_outputs(n, _lib.MODE_PATH, packed_flags=True)
words = first["x_hit"].numel()
spacer_gb = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
M = int(sys.argv[2]) if len(sys.argv) > 2 else 10
pool = [first["x_hit"], first["k_out"]]
spacers = []
for q in range(M - 2):
    spacers.append(torch.empty(int(spacer_gb * 1e9), dtype=torch.uint8, device=dev))
    pool.append(torch.empty(words, dtype=torch.float64, device=dev))
print("free/total GB after allocation:", [round(v / 1e9, 1) for v in torch.cuda.mem_get_info()])
print("array addresses:", " ".join("%x" % t.data_ptr() for t in pool))
sysd.trace_timed(x0, k0, first, 40, e0d)
def march(i, j):
    b = dict(first, x_hit=pool[i], k_out=pool[j])
    sysd.trace_timed(x0, k0, b, 1, e0d)
    return sysd.trace_timed(x0, k0, b, 4, e0d)
for i in range(M):
    print("x%-2d: " % i + " ".join(("%.3f" % march(i, j)) if j != i else "  -  " for j in range(M)), flush=True)
