"""TEST INFRASTRUCTURE: pyrate_amd/engine.py ITSELF -- DeviceSystem.alloc_outputs / trace_into / views / trace / propagate /
interact / surface_step / shape_eval, the layout arithmetic and the argument marshalling the product runs on a GPU box -- on
the HOST build of libprt (tests/hostemu), with CPU tensors.  ``engine_on_host()`` patches, for the duration of a test and
with pytest's MonkeyPatch, the handful of places where engine.py asks torch for a CUDA device or a stream, and hands
``pyrate_amd._lib`` the host build's handle; nothing in the product changes and nothing of this can happen outside a test
(the product refuses the host build: tests/test_hostemu.py).  What it reaches that the C-ABI driver of tests/hostemu does
not: the Python of engine.py, which the build container otherwise never executes."""
import contextlib
import ctypes

import pytest
import torch

from . import load


class _Lib(object):
    """the host build's handle; a device index of None (the index of torch.device("cpu")) is device 0"""
    DEVICE_FIRST = ("prt_rect_grid_count", "prt_collimated_bundle", "prt_raster_count", "prt_raster_bundle", "prt_efield_perp",
                    "prt_poynting_dir", "prt_path_sums", "prt_bundle_moments", "prt_bundle_moments_async")

    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name in self.DEVICE_FIRST:
            return lambda dev, *a: fn(0 if dev is None else dev, *a)
        if name == "prt_trace_seq":
            return lambda *a: fn(*[(0 if (i == 12 and v is None) else v) for (i, v) in enumerate(a)])
        return fn


class _Stream(object):
    cuda_stream = 0

    def synchronize(self):
        pass

    def wait_event(self, *a):
        pass


@contextlib.contextmanager
def engine_on_host():
    from pyrate_amd import engine, placed, _lib as product
    lib = _Lib(load())
    mp = pytest.MonkeyPatch()
    try:
        mp.setattr(product, "_lib", lib)                       # product.load() / check() use the host build's handle
        mp.setattr(torch.cuda, "is_available", lambda: True)
        mp.setattr(torch.cuda, "init", lambda: None)
        mp.setattr(torch.cuda, "device", contextlib.nullcontext)
        mp.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
        mp.setattr(torch.cuda, "current_device", lambda: 0)
        mp.setattr(torch.cuda, "current_stream", lambda *a, **k: _Stream())
        mp.setattr(torch.cuda, "empty_cache", lambda: None)

        def pageable(fn):           # page-locked host buffers need a device runtime: plain host memory here
            return lambda *a, **k: fn(*a, **{key: v for (key, v) in k.items() if key != "pin_memory"})
        mp.setattr(torch, "zeros", pageable(torch.zeros))
        mp.setattr(torch, "empty", pageable(torch.empty))
        mp.setattr(engine, "raw_stream", lambda device: 0)
        mp.setattr(engine, "_current_device", lambda: None)
        mp.setattr(torch.Tensor, "is_cuda", property(lambda self: True), raising=False)
        mp.setattr(placed, "DISABLED", "the host build of the kernels (tests): no placement arena")
        real = engine.DeviceSystem

        class HostEngineSystem(real):
            """engine.DeviceSystem with its tensors on the CPU; every method except the constructor is the product's"""

            def __init__(self, records, device=0):
                from pyrate_amd import surface_table
                self.lib = product.load()
                self.device = torch.device("cpu")
                self.records = list(records)
                self.n_surfaces = len(self.records)
                self._table = engine.pack_table(self.records)
                self.complex_eps = surface_table.has_complex_eps(self.records)
                self.all_isotropic = all(r["material"]["type"] == "isotropic" for r in self.records) and not self.complex_eps
                handle = ctypes.c_void_p()
                product.check(self.lib.prt_system_create(self._table, self.n_surfaces, 0, ctypes.byref(handle)))
                self._h = handle
                self._counts = {}
                self.updates = 0
        mp.setattr(engine, "DeviceSystem", HostEngineSystem)
        from pyrate_amd.raytracer import ray
        mp.setattr(ray, "_DEFAULT_DEVICE", [torch.device("cpu")])        # bundles made from NumPy arrays live on the CPU
        yield engine
    finally:
        mp.undo()
