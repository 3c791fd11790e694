"""Same process, ONE arena: does the march time depend on where x_hit / k_out / flags sit relative
to each other (channel aliasing between the concurrently written rows) or only on the allocation?"""
import sys, torch
sys.path.insert(0, '.')
from pyrate_amd import engine, systems, _lib
dev = torch.device("cuda", 0)
sysd = engine.DeviceSystem(systems.double_gauss_records(), 0)
(x0, k0, e0d, _) = systems.double_gauss_bundle_device(10000000, dev)
n = x0.shape[1]
S = 12
pitch = (n + 511) // 512 * 512
xb = 3 * S * pitch * 8
vb = S * pitch
arena = torch.empty(3 * xb, dtype=torch.uint8, device=dev)
base = arena.data_ptr()
print("arena @%x  pitch*8 = %d = %d * 4096" % (base, pitch * 8, pitch * 8 // 4096))

def run(off_x, off_k, off_v, tag, p=pitch):
    bufs = dict(x_hit=arena[off_x:off_x + 3 * S * p * 8].view(torch.float64),
                k_out=arena[off_k:off_k + 3 * S * p * 8].view(torch.float64),
                valid=arena[off_v:off_v + S * p], valid_out=None,
                n_in=[n] * S, n_out=[n] * S, mode=_lib.MODE_PATH, pitch=p, packed_flags=True)
    sysd.trace_timed(x0, k0, bufs, 3, e0d)
    ms = sysd.trace_timed(x0, k0, bufs, 20, e0d)
    print("%-44s x@+%-12d k@+%-12d : %.4f ms" % (tag, off_x, off_k, ms), flush=True)

run(0, xb, 2 * xb, "warm")
for rep in range(3):
    run(0, xb, 2 * xb, "baseline x | k | flags")
for d in (256, 1024, 4096, 8192, 65536, 1 << 20, (1 << 20) + 4096, 3 << 20):
    run(0, xb + d, 2 * xb + 2 * d, "k shifted by %d" % d)
for sh in (4096, 65536, 1 << 21, 1 << 24, 1 << 28):
    run(sh, xb + sh, 2 * xb + sh, "everything shifted by %d" % sh)
run(0, xb, 2 * xb, "baseline again")
# a second arena: same layout, other allocation
arena2 = torch.empty(3 * xb, dtype=torch.uint8, device=dev)
keep = arena
arena = arena2
print("arena2 @%x" % arena.data_ptr())
for rep in range(2):
    run(0, xb, 2 * xb, "arena2 baseline")
arena = keep
run(0, xb, 2 * xb, "arena1 baseline")
