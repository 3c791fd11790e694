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
            e = None if uniform.e_re is None else np.asarray(uniform.e_re) + 1j * np.asarray(uniform.e_im)
            uni = (uniform.k, e, uniform._first_dir())
        dense = self.hs.trace(host(x0), host(k0), host(e0_re), host(e0_im), mode=mode, in_pitch=in_pitch, uniform=uni,
                              first_dir=first_dir, want_nonconv=want_nonconv and not want_fields, want_fields=want_fields,
                              flags=packed_flags and self.all_isotropic, want_k_im=want_k_im)
        res = _Result()
        res.x_hit = [torch.from_numpy(d["x_hit"]) for d in dense]
        res.k_out = [torch.from_numpy(np.ascontiguousarray(np.real(d["k_out"]))) for d in dense]
        res.valid = [torch.from_numpy(np.ascontiguousarray(d["valid"])) for d in dense]
        res.valid_out = [torch.from_numpy(np.ascontiguousarray(d["valid_out"])) for d in dense]
        res.nonconv = [torch.from_numpy(np.ascontiguousarray(d["nonconv"])) for d in dense] if "nonconv" in dense[0] else None
        res.flags = None
        if packed_flags and self.all_isotropic:
            res.flags = [torch.from_numpy(np.ascontiguousarray(d["valid"] | (d["valid_out"] << 1) | (d["nonconv"] << 2)))
                         for d in dense]
        res.k_out_im = [torch.from_numpy(d["k_im"]) for d in dense] if "k_im" in dense[0] else None
        res.e_out = ([(torch.from_numpy(d["e_re"]), torch.from_numpy(d.get("e_im", np.zeros_like(d["e_re"])))) for d in dense]
                     if "e_re" in dense[0] else None)
        (res.n_in, res.n_out) = self.hs.ray_counts(n0)
        res.mode = mode
        return res

    # -- one surface at a time (the tight one-ray-per-thread entry points and the row entry points) ----------------------
    def propagate(self, surface, x, k, direction=None, e_re=None, e_im=None, default_e=True, valid_in=None,
                  want_nonconv=False, placement="auto"):
        def host(t):
            return None if t is None else np.ascontiguousarray(t.numpy())
        out = self.hs.propagate_rows(surface, host(x), host(k), direction=host(direction), e_re=host(e_re), e_im=host(e_im),
                                     default_e=default_e, valid_in=host(valid_in), want_nonconv=want_nonconv)
        return tuple(torch.from_numpy(a) for a in out)

    def interact(self, surface, x_hit, k, valid_in=None, want_e=False, want_dir=None, placement="auto"):
        if self.records[surface]["material"]["type"] == "anisotropic" or self.complex_eps:
            raise NotImplementedError("adapter: crystal interfaces one at a time")
        out = self.hs.interact_rows(surface, np.ascontiguousarray(x_hit.numpy()), np.ascontiguousarray(k.numpy()),
                                    valid_in=None if valid_in is None else valid_in.numpy(), want_dir=bool(want_dir))
        (k_out, valid_out) = (torch.from_numpy(out[0]), torch.from_numpy(out[1]))
        return k_out, (torch.from_numpy(out[2]) if want_dir else None), valid_out, None, None

    def surface_step(self, surface, x, k, direction=None, e_re=None, e_im=None, default_e=True, valid_in=None,
                     want_nonconv=False, placement="auto"):
        def host(t):
            return None if t is None else np.ascontiguousarray(t.numpy())
        if self.records[surface]["material"]["type"] == "anisotropic" or self.complex_eps:
            raise ValueError("surface_step: isotropic, lossless media only (crystals: propagate + interact)")
        out = self.hs.surface_step_rows(surface, host(x), host(k), direction=host(direction), e_re=host(e_re), e_im=host(e_im),
                                        default_e=default_e, valid_in=host(valid_in), want_nonconv=want_nonconv)
        return tuple(torch.from_numpy(a) for a in out)

    def shape_eval(self, surface, x, y, want_sag=True, want_grad=True):
        (sag, grad) = self.hs.shape_eval(surface, x.numpy(), y.numpy())
        return (torch.from_numpy(sag) if want_sag else None, torch.from_numpy(grad) if want_grad else None)
