"""A/B of libprt builds in ONE process on the SAME input and output arrays (placement is worth
+-10 %, so variants must not get their own allocations): path-mode and image-mode march."""
import ctypes, glob, os, sys, torch
sys.path.insert(0, '.')
from pyrate_amd import engine, systems, _lib
dev = torch.device("cuda", 0)
sysd = engine.DeviceSystem(systems.asphere_records(coefficients=(1e-3, -1e-6, 1e-8), curv=-1. / 30., cc=-1.5), 0)
(x0, k0, e0d, _) = systems.double_gauss_bundle_device(10000000, dev, rpup=9.0, z0=-5.0, field_deg=5.0)
n = x0.shape[1]
(bufs, rep) = sysd.alloc_outputs_tuned(x0, k0, e0d, packed_flags=True)
print("placement: first pair %.4f, chosen %.4f" % (rep["first_pair_ms"], rep["best_pair_ms"]))
# a pair of the slow kind for comparison: fresh arrays until one is found
bad = None
keep = []
for _ in range(12):
    c = sysd.alloc_outputs(n, _lib.MODE_PATH, packed_flags=True)
    keep.append(c)
    sysd.trace_timed(x0, k0, c, 1, e0d)
    t = sysd.trace_timed(x0, k0, c, 3, e0d)
    if t > 0.53:
        bad = c
        print("slow pair found: %.4f" % t)
        break
img = sysd.alloc_outputs(n, _lib.MODE_IMAGE, packed_flags=True)
libs = {"in-tree": (sysd.lib, sysd._h)}
for path in sorted(glob.glob("scratch/variants/libprt_*.so")):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name in ("prt_system_create", "prt_trace_timed", "prt_system_destroy"):
        (res, args) = _lib.PROTOTYPES[name]
        getattr(lib, name).restype = res
        getattr(lib, name).argtypes = args
    h = ctypes.c_void_p()
    assert lib.prt_system_create(sysd._table, sysd.n_surfaces, 0, ctypes.byref(h)) == 0
    libs[os.path.basename(path)[7:-3]] = (lib, h)
st = engine._stream_handle(dev)
P = engine._ptr

def timed(lib, h, b, iters):
    ms = ctypes.c_double()
    rc = lib.prt_trace_timed(h, n, sysd._in_pitch(x0, k0, e0d, None), P(x0), P(k0), P(e0d), None,
                             engine._mode_word(b), b["pitch"], P(b["x_hit"]), P(b["k_out"]), P(b["valid"]),
                             None, st, iters, ctypes.byref(ms))
    assert rc == 0, rc
    return ms.value

timed(*libs["in-tree"], bufs, 40)
for rep in range(3):
    for (tag, b, it) in (("path ", bufs, 20), ("image", img, 30)) + ((("slow ", bad, 20),) if bad else ()):
        print(tag + " " + "  ".join("%s %.4f" % (name, (timed(lib, h, b, 2), timed(lib, h, b, it))[1])
                                    for (name, (lib, h)) in libs.items()), flush=True)
