"""
``Material`` plugin interface of the reference (raytracer/material/material.py:36-75):
``propagate(raybundle, nextSurface)``, ``refract(raybundle, actualSurface, splitup)``,
``reflect(...)``.  The two helpers below are the only places where the per-surface HIP
entry points (prt_propagate / prt_interact / prt_compact) are called for the
plugin-granular API; whole sequences go through ``OpticalSystem.seqtrace`` -> prt_trace.
"""
import torch

from ... import engine
from ..variables import Named
from .. import _dispatch
from ..ray import RayBundle


class _Vacuum(object):
    def __init__(self, lc):
        self.lc = lc

    def get_optical_index(self, x, wave):
        return 1.0


def propagate_bundle(raybundle, shape, aperture):
    """Surface.intersect for a device bundle: appends the hit point, cumulative valid
    (surface.py:116-135, surface_shape.py:289-325 / 448-465, ray.py:83-105)."""
    raybundle._ensure()
    dev = raybundle.device
    sysd = _dispatch.single_surface_system(shape, aperture, _Vacuum(shape.lc), False,
                                           raybundle.wave, dev)
    x = raybundle._x[-1]
    k = raybundle._k[-1]
    (direction, e_re, e_im, default_e) = (None, None, None, False)
    if raybundle._dir is not None:
        direction = raybundle._dir                # rays inside an anisotropic medium
    elif raybundle._dir_from_k:
        # behind an isotropic interface: d = k/|k| -- for a REAL k.  With a complex one (an isotropic medium behind an
        # absorbing one) the reference takes E from an SVD with a two-dimensional null space, and the ray direction
        # with it: LAPACK's arbitrary pick, nothing to be compatible with (include/prt.h, eps_im)
        if getattr(raybundle, "_k_im", None) is not None:
            from ...surface_table import UnsupportedError
            raise UnsupportedError("propagate: a bundle with complex wave vectors that left an isotropic interface has "
                                   "no defined ray direction (complex wave vectors are defined inside crystals and "
                                   "behind the last surface only)")
    else:
        e = raybundle._e[-1]                      # user bundle: Poynting direction of (k, E)
        if e is None:
            default_e = True
        else:
            (e_re, e_im) = e
    (x_hit, valid) = sysd.propagate(0, x, k, direction=direction, e_re=e_re, e_im=e_im,
                                    default_e=default_e, valid_in=raybundle._valid[-1])
    raybundle._append_device(x_hit, valid)


def interact_bundle(material, raybundle, surface, mirror, splitup):
    """Material.refract / reflect for a device bundle -> tuple of new RayBundles"""
    raybundle._ensure()
    dev = raybundle.device
    sysd = _dispatch.single_surface_system(surface.shape, None, material, mirror,
                                           raybundle.wave, dev)
    x_hit = raybundle._x[-1]
    k = raybundle._k[-1]
    aniso = sysd.records[0]["material"]["type"] == "anisotropic"
    ids = raybundle.ray_ids_dev()
    k_im = getattr(raybundle, "_k_im", None)
    if sysd.complex_eps or k_im is not None:
        return _interact_bundle_cplx(sysd, raybundle, x_hit, k, None if k_im is None else k_im[-1], ids, aniso, splitup)
    if not aniso:
        (k_out, _d, valid_out, _, _) = sysd.interact(0, x_hit, k, valid_in=raybundle._valid[-1])
        # return only valid rays (material_isotropic.py:194-199), on the device
        ((xc, kc), idc, _) = engine.compact(valid_out, [x_hit, k_out], ids)
        ones = torch.ones(xc.shape[1], dtype=torch.uint8, device=dev)
        return (RayBundle._from_device([xc], [kc], [ones], idc, raybundle.wave, dev,
                                       dir_from_k=True, k_complex=raybundle._k_complex),)
    (k_out, dir_out, _v, e_re, e_im) = sysd.interact(0, x_hit, k, want_e=True)
    n = x_hit.shape[1]
    if not splitup:
        x2 = torch.cat((x_hit, x_hit), dim=1).contiguous()
        id2 = torch.cat((ids, ids))
        ones = torch.ones(2 * n, dtype=torch.uint8, device=dev)
        return (RayBundle._from_device([x2], [k_out], [ones], id2, raybundle.wave, dev,
                                       e_list=[(e_re, e_im)], direction=dir_out, dir_from_k=False,
                                       k_complex=True, splitted=True),)
    out = []
    for b in range(2):
        sl = slice(b * n, (b + 1) * n)
        ones = torch.ones(n, dtype=torch.uint8, device=dev)
        out.append(RayBundle._from_device(
            [x_hit], [k_out[:, sl].contiguous()], [ones], ids, raybundle.wave, dev,
            e_list=[(e_re[:, sl].contiguous(), e_im[:, sl].contiguous())],
            direction=dir_out[:, sl].contiguous(), dir_from_k=False, k_complex=True))
    return tuple(out)


def _interact_bundle_cplx(sysd, raybundle, x_hit, k, k_im, ids, aniso, splitup):
    """interact_bundle for complex wave vectors (prt_interact_cplx): an absorbing medium behind the surface, or a
    bundle that comes out of one.  Same bundle structure as the real case; the new bundles carry Im(k)."""
    dev = raybundle.device
    wave = raybundle.wave

    def with_k_im(bundle, kim):
        bundle._k_im = [kim]
        return bundle
    if not aniso:
        (k_out, k_out_im, _d, valid_out, _, _) = sysd.interact_cplx(0, x_hit, k, k_im, valid_in=raybundle._valid[-1])
        ((xc, kc, kic), idc, _) = engine.compact(valid_out, [x_hit, k_out, k_out_im], ids)
        ones = torch.ones(xc.shape[1], dtype=torch.uint8, device=dev)
        return (with_k_im(RayBundle._from_device([xc], [kc], [ones], idc, wave, dev, dir_from_k=True, k_complex=True),
                          kic),)
    (k_out, k_out_im, dir_out, _v, e_re, e_im) = sysd.interact_cplx(0, x_hit, k, k_im, want_e=True)
    n = x_hit.shape[1]
    if not splitup:
        x2 = torch.cat((x_hit, x_hit), dim=1).contiguous()
        id2 = torch.cat((ids, ids))
        ones = torch.ones(2 * n, dtype=torch.uint8, device=dev)
        return (with_k_im(RayBundle._from_device([x2], [k_out], [ones], id2, wave, dev, e_list=[(e_re, e_im)],
                                                 direction=dir_out, dir_from_k=False, k_complex=True, splitted=True),
                          k_out_im),)
    out = []
    for b in range(2):
        sl = slice(b * n, (b + 1) * n)
        ones = torch.ones(n, dtype=torch.uint8, device=dev)
        out.append(with_k_im(RayBundle._from_device(
            [x_hit], [k_out[:, sl].contiguous()], [ones], ids, wave, dev,
            e_list=[(e_re[:, sl].contiguous(), e_im[:, sl].contiguous())],
            direction=dir_out[:, sl].contiguous(), dir_from_k=False, k_complex=True), k_out_im[:, sl].contiguous()))
    return tuple(out)


class Material(Named):
    """abstract base class for materials"""
    kind = "material"

    def __init__(self, lc, name="", comment=""):
        Named.__init__(self, name)
        self.lc = lc
        self.annotations["comment"] = comment
        self.comment = comment

    @classmethod
    def p(cls, lc, name="", comment=""):
        return cls(lc, name=name, comment=comment)

    def propagate(self, raybundle, nextSurface):
        """propagates the bundle to nextSurface: appends the intersection point (mutates
        the bundle)"""
        nextSurface.intersect(raybundle)

    def refract(self, raybundle, actualSurface, splitup=False):
        return interact_bundle(self, raybundle, actualSurface, False, splitup)

    def reflect(self, raybundle, actualSurface, splitup=False):
        return interact_bundle(self, raybundle, actualSurface, True, splitup)
