"""does a ray's result depend on whether it is the first or the second ray of its thread?  trace the
bundle and the bundle shifted by one ray; compare the overlap bit for bit (in-tree build and variants)"""
import ctypes, glob, os, sys, torch
sys.path.insert(0, '.')
from pyrate_amd import engine, systems, _lib
dev = torch.device("cuda", 0)
def same(a, b):
    return torch.equal(a.contiguous().view(torch.int64), b.contiguous().view(torch.int64))
libs = [None] + sorted(glob.glob("scratch/variants/libprt_*.so"))
for path in libs:
    if path:
        os.environ["PRT_LIBRARY"] = os.path.abspath(path)
    # fresh interpreter state per library is simpler: re-exec
import subprocess
code = r'''
import sys, torch
sys.path.insert(0, '.')
from pyrate_amd import engine, systems, _lib
dev = torch.device("cuda", 0)
def same(a, b):
    return torch.equal(a.contiguous().view(torch.int64), b.contiguous().view(torch.int64))
for (name, recs) in (("double gauss", systems.double_gauss_records()), ("asphere", systems.asphere_records(coefficients=(1e-3, -1e-6, 1e-8), curv=-1. / 30., cc=-1.5))):
    sysd = engine.DeviceSystem(recs, 0)
    (x0, k0, e0d, n) = systems.double_gauss_bundle_device(2000000, dev, field_deg=2.0)
    a = sysd.trace(x0, k0, e0d, mode=_lib.MODE_IMAGE)
    b = sysd.trace(x0[:, 1:].contiguous(), k0[:, 1:].contiguous(), e0d[:, 1:].contiguous(), mode=_lib.MODE_IMAGE)
    d = (a.x_hit[0][:, 1:] - b.x_hit[0]).abs().max(dim=0).values
    print("   %-12s shifted-by-one identical: x %s k %s  (rays that differ: %d of %d, max %.2e)" % (name, same(a.x_hit[0][:, 1:], b.x_hit[0]), same(a.k_out[0][:, 1:], b.k_out[0]), int((d > 0).sum()), d.numel(), float(d.max())))
'''
for path in libs:
    env = dict(os.environ)
    if path:
        env["PRT_LIBRARY"] = os.path.abspath(path)
    else:
        env.pop("PRT_LIBRARY", None)
    print("library:", path or "in-tree", flush=True)
    subprocess.run([sys.executable, "-c", code], env=env)
