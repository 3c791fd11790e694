"""TEST INFRASTRUCTURE: the few members of ``pyrate_amd.engine`` that the bodies of the `-m gpu` tests use
(``DeviceSystem(recs, 0).trace(...)``, ``to_device_rays``), implemented on the HOST build of libprt (tests/hostemu), with
torch CPU tensors where the engine hands out device tensors.  ``tests/test_hostemu_campaigns.py`` swaps them into
``pyrate_amd.engine`` with pytest's monkeypatch for the duration of one test and calls the body of the GPU test -- so the
comparison that runs on the device at round end is the comparison that runs here.  The product never sees this module."""
import numpy as np
import torch

from . import HostSystem


class _Result(object):
    pass


def to_device_rays(a, device=None, pitched=True):
    a = np.asarray(a)
    if np.iscomplexobj(a):
        a = a.real
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))
    t.prt_pitched = bool(pitched)
    return t


class HostDeviceSystem(object):
    def __init__(self, records, device=0):
        self.hs = HostSystem(records)
        self.records = self.hs.records
        self.n_surfaces = self.hs.S
        self.complex_eps = self.hs.complex_eps
        self.all_isotropic = self.hs.all_isotropic

    def ray_counts(self, n0):
        return self.hs.ray_counts(n0)

    def trace(self, x0, k0, e0_re=None, e0_im=None, mode=0, want_fields=False, packed_flags=False, want_nonconv=False,
              uniform=None, first_dir=None, want_k_im=False):
        def host(t):
            return None if t is None else t.numpy()
        n0 = x0.shape[1]
        # pitched inputs (to_device_rays' default) -> rows of the recommended pitch, as on the device
        in_pitch = int(self.hs.lib.prt_recommended_pitch(n0)) if getattr(x0, "prt_pitched", False) and n0 else None
        uni = None
        if uniform is not None:
            e = uniform.e
            uni = (uniform.k, e, {"e": 3, "k": 1, "dir": 4}[uniform.kind] if e is not None or uniform.kind != "e" else 0)
        dense = self.hs.trace(host(x0), host(k0), host(e0_re), host(e0_im), mode=mode, in_pitch=in_pitch, uniform=uni,
                              first_dir=first_dir, want_nonconv=want_nonconv and not want_fields, want_fields=want_fields,
                              flags=packed_flags and self.all_isotropic, want_k_im=want_k_im)
        res = _Result()
        res.x_hit = [torch.from_numpy(d["x_hit"]) for d in dense]
        res.k_out = [torch.from_numpy(np.ascontiguousarray(np.real(d["k_out"]))) for d in dense]
        res.valid = [torch.from_numpy(np.ascontiguousarray(d["valid"])) for d in dense]
        res.valid_out = [torch.from_numpy(np.ascontiguousarray(d["valid_out"])) for d in dense]
        res.nonconv = [torch.from_numpy(np.ascontiguousarray(d["nonconv"])) for d in dense] if "nonconv" in dense[0] else None
        res.k_out_im = [torch.from_numpy(d["k_im"]) for d in dense] if "k_im" in dense[0] else None
        res.e_out = ([(torch.from_numpy(d["e_re"]), torch.from_numpy(d.get("e_im", np.zeros_like(d["e_re"])))) for d in dense]
                     if "e_re" in dense[0] else None)
        (res.n_in, res.n_out) = self.hs.ray_counts(n0)
        res.mode = mode
        return res
