"""march time per output allocation: separate x/k/flags allocations vs one arena each; same process"""
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
sets = []
for q in range(6):
    sets.append(("separate %d" % q, sysd.alloc_outputs(n, _lib.MODE_PATH, packed_flags=True)))
for q in range(6):
    arena = torch.empty(2 * xb + S * pitch, dtype=torch.uint8, device=dev)
    sets.append(("arena %d" % q, dict(x_hit=arena[0:xb].view(torch.float64), k_out=arena[xb:2 * xb].view(torch.float64),
                                      valid=arena[2 * xb:], valid_out=None, n_in=[n] * S, n_out=[n] * S,
                                      mode=_lib.MODE_PATH, pitch=pitch, packed_flags=True)))
sysd.trace_timed(x0, k0, sets[0][1], 40, e0d)
for rep in range(3):
    out = []
    for (tag, b) in sets:
        sysd.trace_timed(x0, k0, b, 3, e0d)
        out.append("%.3f" % sysd.trace_timed(x0, k0, b, 15, e0d))
    print("rep %d: separate " % rep + " ".join(out[:6]) + " | arena " + " ".join(out[6:]), flush=True)
for (tag, b) in sets:
    print(tag, "x@%x k@%x v@%x" % (b["x_hit"].data_ptr(), b["k_out"].data_ptr(), b["valid"].data_ptr()))
