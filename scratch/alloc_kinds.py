"""march time with output arrays from different allocation APIs / flags (raw pointers into prt_trace_timed)"""
import ctypes, sys, torch
sys.path.insert(0, '.')
from pyrate_amd import engine, systems, _lib
dev = torch.device("cuda", 0)
sysd = engine.DeviceSystem(systems.double_gauss_records(), 0)
(x0, k0, e0d, _) = systems.double_gauss_bundle_device(10000000, dev)
n = x0.shape[1]
S = 12
pitch = (n + 511) // 512 * 512
xb = 3 * S * pitch * 8
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipFree.argtypes = [ctypes.c_void_p]
flags_buf = torch.empty(S * pitch, dtype=torch.uint8, device=dev)
st = engine._stream_handle(dev)
P = engine._ptr
lib = sysd.lib

def alloc(flag):
    p = ctypes.c_void_p()
    rc = hip.hipMalloc(ctypes.byref(p), xb) if flag is None else hip.hipExtMallocWithFlags(ctypes.byref(p), xb, flag)
    return p.value if rc == 0 else None

def timed(px, pk, iters):
    ms = ctypes.c_double()
    rc = lib.prt_trace_timed(sysd._h, n, sysd._in_pitch(x0, k0, e0d, None), P(x0), P(k0), P(e0d), None,
                             _lib.MODE_PATH | _lib.MODE_FLAGS, pitch, px, pk, P(flags_buf), None, st, iters,
                             ctypes.byref(ms))
    assert rc == 0, (rc, lib.prt_last_error())
    return ms.value

warm = sysd.alloc_outputs(n, _lib.MODE_PATH, packed_flags=True)
sysd.trace_timed(x0, k0, warm, 40, e0d)
print("torch arrays: %.4f" % sysd.trace_timed(x0, k0, warm, 5, e0d))
for (name, flag) in (("hipMalloc", None), ("ext default", 0), ("ext finegrained", 1), ("ext uncached", 3), ("ext contiguous", 4)):
    ptrs = [alloc(flag) for _ in range(4)]
    if any(p is None for p in ptrs):
        print(name, "allocation failed"); continue
    res = []
    for (i, j) in ((0, 1), (2, 3), (0, 2), (1, 3)):
        timed(ptrs[i], ptrs[j], 1)
        res.append("%.4f" % timed(ptrs[i], ptrs[j], 4))
    # mixed with a torch array
    timed(P(warm["x_hit"]), ptrs[0], 1)
    mix = timed(P(warm["x_hit"]), ptrs[0], 4)
    print("%-16s pairs %s | torch x + this k %.4f" % (name, " ".join(res), mix), flush=True)
    for p in ptrs:
        hip.hipFree(p)
