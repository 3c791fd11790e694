"""one arena: march time vs the offset between x_hit and k_out (large shifts: 2 MiB ... GiBs)"""
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
MiB = 1 << 20
arena = torch.empty(2 * xb + 20 * 1024 * MiB, dtype=torch.uint8, device=dev)
flags = torch.empty(S * pitch, dtype=torch.uint8, device=dev)
print("arena @%x, x_hit bytes %d = %.3f MiB" % (arena.data_ptr(), xb, xb / MiB))
def run(off_x, off_k):
    b = dict(x_hit=arena[off_x:off_x + xb].view(torch.float64), k_out=arena[off_k:off_k + xb].view(torch.float64),
             valid=flags, valid_out=None, n_in=[n] * S, n_out=[n] * S, mode=_lib.MODE_PATH, pitch=pitch,
             packed_flags=True)
    sysd.trace_timed(x0, k0, b, 1, e0d)
    return sysd.trace_timed(x0, k0, b, 4, e0d)
run(0, xb); sysd.trace_timed(x0, k0, dict(x_hit=arena[0:xb].view(torch.float64), k_out=arena[xb:2*xb].view(torch.float64), valid=flags, valid_out=None, n_in=[n]*S, n_out=[n]*S, mode=0, pitch=pitch, packed_flags=True), 40, e0d)
base_k = (xb + 2 * MiB - 1) // (2 * MiB) * (2 * MiB)          # k_out on a 2-MiB boundary right behind x_hit
print("k directly behind x (unaligned): %.4f   k at next 2-MiB boundary: %.4f" % (run(0, xb), run(0, base_k)))
for mult in (1, 2, 3, 4, 5, 6, 7, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096, 6144, 8192):
    d = mult * 2 * MiB
    print("k shifted by %5d x 2 MiB (%8.1f MiB): %.4f" % (mult, d / MiB, run(0, base_k + d)), flush=True)
print("x shifted too (x at +1 GiB, k right behind): %.4f" % run(1024 * MiB, 1024 * MiB + base_k))
