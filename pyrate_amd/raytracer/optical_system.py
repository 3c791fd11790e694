"""
``OpticalSystem`` with the reference's ``seqtrace`` signature and return structure
(raytracer/optical_system.py:42-94).

``seqtrace(initialbundle, elementsequence, splitup=False) -> list[RayPath]``:
* all-isotropic sequence: the whole sequence is flattened into one surface table and
  traced by ONE fused HIP launch (prt_trace, path mode).  The returned RayBundles are
  lazily compacted views into the dense device arrays (see ray.py) and reproduce the
  reference's bundle structure, including the bundle that appears twice at every element
  boundary (optical_element.py:330, ray.py:218-219).
* sequences through anisotropic media (ray doubling, ``splitup`` forking, E fields):
  the reference's own element / surface loops run on top of the per-surface HIP entry
  points (optical_element.py).
"""
import torch

from .. import _lib, engine
from ..surface_table import flatten_sequence
from . import _dispatch
from .localcoordinates import LocalCoordinates, LocalCoordinatesTreeBase
from .material.material_isotropic import ConstantIndexGlass
from .ray import RayBundle, RayPath


class OpticalSystem(LocalCoordinatesTreeBase):
    kind = "opticalsystem"

    def __init__(self, rootlc, matbackground, name=""):
        LocalCoordinatesTreeBase.__init__(self, rootlc, name=name)
        self.material_background = matbackground
        self.elements = {}

    @classmethod
    def p(cls, rootlc=None, matbackground=None, name=""):
        if rootlc is None:
            rootlc = LocalCoordinates.p(name="global")
        if matbackground is None:
            matbackground = ConstantIndexGlass.p(rootlc, 1.0, name="background")
        return cls(rootlc, matbackground, name=name)

    def addElement(self, key, element):
        if self.checkForRootConnection(element.rootcoordinatesystem):
            self.elements[key] = element
        else:
            raise Exception("OpticalElement root should be connected to root of OpticalSystem")

    def removeElement(self, key):
        if key in self.elements:
            self.elements.pop(key)

    # ------------------------------------------------------------------
    def seqtrace(self, initialbundle, elementsequence, splitup=False):
        # e.g. [("elem1", [("surf1", {}), ("surf2", {"is_mirror": True})]), ("elem2", [...])]
        (records, lengths) = flatten_sequence(self, elementsequence, initialbundle.wave)
        initialbundle._ensure()
        fused_ok = all(r["material"]["type"] == "isotropic" for r in records) \
            and initialbundle._dir is None and len(records) > 0
        if fused_ok and len(initialbundle._valid) > 1 and not bool(initialbundle._valid[-1].all()):
            fused_ok = False      # a bundle that already carries invalid rays: per-surface path
        if fused_ok:
            return [self._seqtrace_fused(initialbundle, records, lengths)]
        return self._seqtrace_generic(initialbundle, elementsequence, splitup)

    def _seqtrace_generic(self, initialbundle, elementsequence, splitup):
        rpaths = [RayPath(initialbundle.clone())]      # do not modify initialbundle (:74)
        for (elem, subseq) in elementsequence:
            rpaths_new = []
            for rp in rpaths:
                to_append = self.elements[elem].seqtrace(rp.raybundles[-1], subseq,
                                                         self.material_background, splitup=splitup)
                for rp_append in to_append[1:]:
                    rpathprime = rp.clone()
                    rpathprime.appendRayPath(rp_append)
                    rpaths_new.append(rpathprime)
                rp.appendRayPath(to_append[0])
            rpaths = rpaths + rpaths_new
        return rpaths

    def _seqtrace_fused(self, ib, records, lengths):
        return seqtrace_fused(ib, records, lengths)


def seqtrace_fused(ib, records, lengths):
    """One fused launch for an all-isotropic flattened sequence -> RayPath with the reference's
    bundle structure (lazy, device-resident).  Needs only the surface-table records, so it
    serves any object graph ``flatten_sequence`` understands (this package's classes or real
    pyrateoptics objects, see pyrate_amd/dropin.py)."""
    dev = ib.device
    sysd = _dispatch.system_for(records, dev)
    S = len(records)
    x0 = ib._x[-1]
    k0 = ib._k[-1]
    (e_re, e_im) = (None, None)
    if ib._dir_from_k:
        e_re = engine.efield_perp(k0)          # E perpendicular to k: Poynting direction = k/|k|
    elif ib._e[-1] is not None:
        (e_re, e_im) = ib._e[-1]
    res = sysd.trace(x0, k0, e_re, e_im, mode=_lib.MODE_PATH)
    n = x0.shape[1]
    if ib._ray_id is None:
        ids0 = torch.arange(n, dtype=torch.int64, device=dev)
    elif isinstance(ib._ray_id, torch.Tensor):
        ids0 = ib._ray_id.to(dev)
    else:
        import numpy as np
        ids0 = torch.from_numpy(np.ascontiguousarray(ib._ray_id, dtype=np.int64)).to(dev)
    wave = ib.wave
    kc = ib._k_complex

    # bundle 0: copy of the initial bundle + the first hit point
    b0 = ib.clone()
    b0._append_device(res.x_hit[0], res.valid[0] * ib._valid[-1])

    def make_thunk(j):
        def thunk(b):
            mask = res.valid_out[j - 1]
            arrays = [res.x_hit[j - 1], res.k_out[j - 1]]
            flags = None
            if j < S:
                arrays.append(res.x_hit[j])
                flags = res.valid[j]
            out = engine.compact(mask, arrays, ids0, flags)
            (cx, ck) = (out[0][0], out[0][1])
            m = cx.shape[1]
            ones = torch.ones(m, dtype=torch.uint8, device=dev)
            b._x = [cx]
            b._k = [ck]
            b._valid = [ones]
            b._e = [None]
            if j < S:
                b._x.append(out[0][2])
                b._k.append(ck)
                b._valid.append(out[2])
                b._e.append(None)
            b._ray_id = out[1]
            b._n = m
            b._k_complex = kc
        return thunk

    bundles = [b0] + [RayBundle._lazy(make_thunk(j), wave, dev) for j in range(1, S + 1)]
    path = RayPath(bundles[0])
    idx = 0
    for L in lengths:
        path.appendRayBundle(bundles[idx])           # the element restarts with the same bundle
        for l in range(L):
            path.appendRayBundle(bundles[idx + l + 1])
        idx += L
    path.dense = res                                  # dense device arrays for GPU consumers
    return path
