"""
``OpticalSystem`` with the reference's ``seqtrace`` signature and return structure
(raytracer/optical_system.py:42-94).

``seqtrace(initialbundle, elementsequence, splitup=False) -> list[RayPath]``:
* the whole sequence is flattened into one surface table and traced by ONE engine call
  (prt_trace, path mode; prt_trace_fields when the sequence crosses anisotropic media, whose
  interfaces double the ray count).  The returned RayBundles are lazily compacted views into
  the dense device arrays (see ray.py) and reproduce the reference's bundle structure,
  including the bundle that appears twice at every element boundary
  (optical_element.py:330, ray.py:218-219).
* ``splitup=True`` through anisotropic media (one RayPath per branch) and bundles that
  already carry invalid rays: the reference's own element / surface loops run on top of the
  per-surface HIP entry points (optical_element.py).
"""
import torch

from .. import _lib, engine
from ..surface_table import UNTRACKED_READS, UnsupportedError, flatten_sequence, has_complex_eps
from . import _dispatch
from .localcoordinates import LocalCoordinates, LocalCoordinatesTreeBase
from .material.material_isotropic import ConstantIndexGlass
from .ray import RayBundle, RayPath
from .variables import mutation_epoch


class _Records(list):
    """the flattened records of a sequence + the facts the dispatch needs about them (computed once per walk)"""
    __slots__ = ("crystals", "complex_eps", "device_systems")

    def __init__(self, records):
        list.__init__(self, records)
        self.device_systems = {}      # device index -> (DeviceSystem, its update count): raytracer/_dispatch.py
        self.crystals = sum(r["material"]["type"] == "anisotropic" for r in records)
        self.complex_eps = has_complex_eps(records)


def _facts(records):
    if isinstance(records, _Records):
        return records.crystals, records.complex_eps
    return (sum(r["material"]["type"] == "anisotropic" for r in records), has_complex_eps(records))


class OpticalSystem(LocalCoordinatesTreeBase):
    kind = "opticalsystem"

    def __init__(self, rootlc, matbackground, name=""):
        LocalCoordinatesTreeBase.__init__(self, rootlc, name=name)
        self.material_background = matbackground
        self.elements = {}
        object.__setattr__(self, "_last_flat", None)     # (not a mutation of the system: OpticalSystem._flattened)

    def _flattened(self, elementsequence, wave):
        """``flatten_sequence`` of this system -- or its result of the last call, when NOTHING has been mutated since
        (``variables.mutation_epoch``: every mutation path of this package's classes advances it) and the sequence
        is equal, entry by entry, to the one of that call (a copy is kept: the caller's list is his to edit).  The
        check costs a comparison of two short nested lists; repeated traces of an unchanged system (field points,
        the merit evaluations between two optimiser steps) skip the walk through the object graph altogether.
        After a mutation the walk itself is cheap where nothing changed (``surface_table.surface_record_cached``)."""
        last = self._last_flat
        now = mutation_epoch()
        if last is not None and last[0] == now and last[1] == wave and last[2] == elementsequence:
            return last[3], last[4]
        untracked = UNTRACKED_READS[0]
        (records, lengths) = flatten_sequence(self, elementsequence, wave)
        records = _Records(records)
        try:
            saved = [(ek, [(sk, dict(opts)) for (sk, opts) in sub]) for (ek, sub) in elementsequence]
        except (TypeError, ValueError):
            saved = None
        if UNTRACKED_READS[0] != untracked:
            saved = None             # a record came from an object without mutation epochs: nothing to vouch for it
        object.__setattr__(self, "_last_flat", None if saved is None else (mutation_epoch(), wave, saved, records, lengths))
        return records, lengths

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
        (records, lengths) = self._flattened(elementsequence, initialbundle.wave)
        initialbundle._ensure()
        (crystals, complex_media) = _facts(records)
        fused_ok = initialbundle._dir is None and len(records) > 0 \
            and (crystals == 0 or (not splitup and crystals <= MAX_FUSED_CRYSTALS))
        if fused_ok and len(initialbundle._valid) > 1 and not bool(initialbundle._valid[-1].all()):
            fused_ok = False      # a bundle that already carries invalid rays: per-surface path
        if fused_ok:
            return [self._seqtrace_fused(initialbundle, records, lengths)]
        if complex_media and splitup and crystals <= MAX_FUSED_CRYSTALS and initialbundle._dir is None \
                and (len(initialbundle._valid) == 1 or bool(initialbundle._valid[-1].all())):
            # absorbing media: the forked paths of ``splitup`` are the branches of ONE dense trace
            return _seqtrace_fused_crystal(initialbundle, records, lengths, split=True)
        # everything else -- explicit first directions, a bundle that already carries invalid rays, more crystal
        # interfaces than the dense arrays hold -- goes through the plugin-granular loop (complex wave vectors
        # included: prt_interact_cplx)
        return self._seqtrace_generic(initialbundle, elementsequence, splitup)

    def image_moments(self, initialbundle, elementsequence):
        """Extension (not in the reference): moments of the image-plane points of an all-isotropic
        sequence WITHOUT materialising the ray path -- one image-mode launch that also reduces
        {count, sum v, sum v*v}, v = image point - vertex of the last surface, over the rays that
        arrive (prt_trace_moments); 7 doubles cross PCIe.  Returns (moments (7,), reference point (3,)).
        ``engine.spot_from_moments`` turns them into RayBundleAnalysis' centroid / RMS spot size; merit
        functions of optimiser loops can use the sums directly."""
        (records, _) = self._flattened(elementsequence, initialbundle.wave)
        if any(r["material"]["type"] != "isotropic" for r in records):
            raise Exception("image_moments: isotropic sequences only")
        initialbundle._ensure()
        dev = initialbundle.device
        sysd = _dispatch.system_for(records, dev)
        x0 = initialbundle._x[-1]
        first = _first_segment(initialbundle)
        n = x0.shape[1]
        key = (dev.index, n)
        cache = OpticalSystem._moment_buffers
        if key not in cache:
            if len(cache) > 8:
                cache.clear()
            cache[key] = (sysd.alloc_outputs(n, _lib.MODE_IMAGE, packed_flags=True),
                          engine.MomentsWorkspace(dev, n_results=1, n_rays=n, host_results=True))
        (bufs, ws) = cache[key]
        sysd.trace_moments_into(x0, first.pop("k0"), bufs, ws, slot=0, **first)
        # the seven doubles were written to page-locked host memory by the reduction kernel itself
        torch.cuda.current_stream(dev).synchronize()
        return (ws.host[0].copy(), sysd.moments_reference())

    _moment_buffers = {}

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


MAX_FUSED_CRYSTALS = 6      # 2**6 rays per initial ray in the dense path arrays


def seqtrace_fused(ib, records, lengths):
    """One engine trace (prt_trace / prt_trace_fields) for a flattened sequence -> RayPath with
    the reference's bundle structure (lazy, device-resident).  Needs only the surface-table
    records, so it serves any object graph ``flatten_sequence`` understands (this package's
    classes or real pyrateoptics objects, see pyrate_amd/dropin.py)."""
    (crystals, complex_media) = _facts(records)
    if crystals or complex_media:
        return _seqtrace_fused_crystal(ib, records, lengths)
    dev = ib.device
    sysd = _dispatch.system_for(records, dev)
    S = len(records)
    x0 = ib._x[-1]
    res = sysd.trace(x0, mode=_lib.MODE_PATH, packed_flags=True, **_first_segment(ib))
    ids_cache = []            # rayIDs on the device: built when the first bundle is touched, not per trace

    def ids0():
        if not ids_cache:
            ids_cache.append(ib.ray_ids_dev())
        return ids_cache[0]
    wave = ib.wave
    kc = ib._k_complex

    c0 = ib.clone()           # taken now: the caller may go on using (and appending to) ``ib``

    def make_thunk(j):
        def thunk(b):
            mask = res.valid_out[j - 1]
            arrays = [res.x_hit[j - 1], res.k_out[j - 1]]
            flags = None
            if j < S:
                arrays.append(res.x_hit[j])
                flags = res.valid[j]
            out = engine.compact(mask, arrays, ids0(), flags)
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

    def bundles():
        return _bundle_list([_first_bundle(c0, res)] + [RayBundle._lazy(make_thunk(j), wave, dev)
                                                        for j in range(1, S + 1)], lengths)
    path = RayPath._deferred(bundles)
    path.dense = res                                  # dense device arrays for GPU consumers
    return path


class _LazyFields(object):
    """(E_re, E_im) of a bundle behind a crystal interface, made when somebody looks: ``pair[0]``, ``pair[1]``"""

    def __init__(self, make):
        self._make = make
        self._pair = None

    def __getitem__(self, i):
        if self._pair is None:
            (er, ei) = self._make()
            self._pair = (er, ei)
            self._make = None
        return self._pair[i]


def _first_bundle(c0, res):
    """bundle 0 of a traced path: ``c0`` -- a copy of the initial bundle, taken by the caller at trace time (the user
    may go on using the original) -- + the first hit point.  The mask arithmetic of the appended point (two small
    kernels) waits until somebody looks at the bundle, like the compaction of the later bundles"""
    def thunk(b):
        c0._append_device(res.x_hit[0], res.valid[0] * c0._valid[-1])
        b.__dict__.update(c0.__dict__)
    return RayBundle._lazy(thunk, c0.wave, c0.device, splitted=c0.splitted)


def _bundle_list(bundles, lengths):
    """the reference's list for a path through elements of ``lengths`` surfaces: initial bundle, then per element
    the bundle it starts from once more (the element restarts with the same bundle) and one bundle per surface"""
    out = [bundles[0]]
    idx = 0
    for L in lengths:
        out.append(bundles[idx])
        out.extend(bundles[idx + 1:idx + L + 1])
        idx += L
    return out


def _assemble_path(bundles, lengths, res):
    path = RayPath()
    path.raybundles = _bundle_list(bundles, lengths)
    path.dense = res                                  # dense device arrays for GPU consumers
    return path


def _first_segment(ib):
    """keyword arguments of DeviceSystem.trace / trace_moments_into that describe the first segment of a
    bundle: a uniform (k, E) for collimated bundles -- only x0 is read --, else the arrays"""
    if getattr(ib, "_uniform", None) is not None:
        return dict(k0=None, uniform=ib._uniform)
    k0 = ib._k[-1]
    if ib._dir_from_k:            # E perpendicular to k: Poynting direction = k/|k|
        return dict(k0=k0, first_dir=_lib.FIRST_K)
    if ib._e[-1] is not None:
        return dict(k0=k0, e0_re=ib._e[-1][0], e0_im=ib._e[-1][1])
    return dict(k0=k0)


def _seqtrace_fused_crystal(ib, records, lengths, split=False):
    """``split=True`` (absorbing crystals with ``splitup``): the list of the 2^A forked paths instead of one path --
    path p follows solution bit j of p at the j-th crystal interface (the order OpticalElement.seqtrace builds:
    existing paths take the first solution, the copies with the second one are appended, optical_element.py:360-375),
    which is branch p mod 2^level of the dense arrays at every level; its bundles hold N rays each.

    Sequences through anisotropic media without ``splitup``: one engine trace in the
    concatenated dense layout (ray count doubles behind every crystal interface, [sol2, sol3]
    stacking), E of the doubled rays from prt_trace_fields.  Bundle j is carved lazily out of
    the dense arrays of surface j-1: the rays the reference still carries there are the dense
    slots with ``valid_out`` set (behind a crystal interface: every ray that was not compacted
    away earlier, since AnisotropicMaterial.refract keeps all rays and restarts validity,
    material_anisotropic.py:91-99; behind an isotropic one: the valid rays,
    material_isotropic.py:194-199)."""
    dev = ib.device
    sysd = _dispatch.system_for(records, dev)
    S = len(records)
    x0 = ib._x[-1]
    first = _first_segment(ib)
    # The E fields of the doubled rays (material_anisotropic.py:91-99) are what RayBundle.Efield shows and what a
    # caller needs who goes on from one of these bundles by hand -- not what the trace needs: ray directions come
    # from closed forms (prt_aniso.h).  Lossless crystals are therefore traced WITHOUT fields (no eigenvectors, a
    # third of the time), and the first look at an E field traces the sequence once more with them
    # (``dense_fields``; the table is looked up again by content: the device system of this call may have been
    # overwritten for another table in the meantime).  Absorbing media keep the fields: their march is per surface.
    eager_fields = sysd.complex_eps
    res = sysd.trace(x0, mode=_lib.MODE_PATH, want_fields=eager_fields, want_k_im=not eager_fields, **first)
    fields_cache = []

    def dense_fields():
        if eager_fields:
            return res.padded
        if not fields_cache:
            again = _dispatch.system_for(records, dev).trace(x0, mode=_lib.MODE_PATH, want_fields=True, **first)
            fields_cache.append(again.padded)
        return fields_cache[0]
    n = x0.shape[1]
    wave = ib.wave
    crystal = [r["material"]["type"] == "anisotropic" for r in records]
    # first surface behind which the wave vectors are complex numbers in the reference (a crystal interface; the
    # absorbing isotropic medium behind the last surface)
    first_crystal = crystal.index(True) if any(crystal) else S - 1
    absorbing = sysd.complex_eps
    # The dense arrays have a ray pitch P >= n (rows of every level on 128-B lines, engine.alloc_outputs): a branch
    # is P slots of which the first n are rays and the rest carry mask 0 -- so the compaction by ``valid_out`` that
    # carves a bundle out of them drops the padding by itself, and nothing has to be gathered beforehand.
    dense = res.padded
    P = res.ray_pitch
    ids0 = ib.ray_ids_dev()
    if P != n:
        ids0 = torch.cat((ids0, torch.zeros(P - n, dtype=ids0.dtype, device=dev)))
    ids_cache = {0: ids0}

    def dense_ids(level):
        # ids of the dense slots after ``level`` doublings: [ids, ids] per crystal interface
        if level not in ids_cache:
            prev = dense_ids(level - 1)
            ids_cache[level] = torch.cat((prev, prev))
        return ids_cache[level]

    def make_thunk(j, branch=None):
        s = j - 1
        level = sum(crystal[:s + 1])
        level_in = sum(crystal[:s])

        def cut(t, lvl):
            # the slots of one branch of a dense array at doubling level lvl (all of it without a branch)
            if branch is None:
                return t
            lo = (branch % (1 << lvl)) * P
            return t[..., lo:lo + P]

        def thunk(b):
            mask = cut(dense.valid_out[s], level)
            xs = cut(dense.x_hit[s], level_in)
            if crystal[s] and branch is None:
                xs = torch.cat((xs, xs), dim=1)
            arrays = [xs, cut(dense.k_out[s], level)]
            with_kim = dense.k_out_im is not None and (crystal[s] or (absorbing and s == S - 1))
            if crystal[s] and eager_fields:
                arrays += [cut(t, level) for t in dense.e_out[s]]
            if with_kim:
                arrays.append(cut(dense.k_out_im[s], level))   # Im(k): evanescent modes / absorbing crystals
            flags = None
            if j < S:
                arrays.append(cut(dense.x_hit[j], level))
                flags = cut(dense.valid[j], level)
            out = engine.compact(mask, arrays, dense_ids(level if branch is None else 0), flags)
            arr = out[0]
            (cx, ck) = (arr[0], arr[1])
            n_e = 2 if (crystal[s] and eager_fields) else 0
            if n_e:
                e = (arr[2], arr[3])
            elif crystal[s]:
                e = _LazyFields(lambda: engine.compact(mask, [cut(t, level) for t in dense_fields().e_out[s]])[0])
            else:
                e = None
            b._k_im = [arr[2 + n_e]] if with_kim else None
            m = cx.shape[1]
            b._x = [cx]
            b._k = [ck]
            b._valid = [torch.ones(m, dtype=torch.uint8, device=dev)]
            b._e = [e]
            if j < S:
                b._x.append(arr[-1])
                b._k.append(ck)
                b._valid.append(out[2])
                b._e.append(e)
                if b._k_im is not None:
                    b._k_im.append(b._k_im[0])
            b._ray_id = out[1]
            b._n = m
            b._k_complex = True if s >= first_crystal else ib._k_complex
        return thunk

    def path_of(branch):
        bundles = [_first_bundle(ib.clone(), res)]
        for j in range(1, S + 1):
            b = RayBundle._lazy(make_thunk(j, branch), wave, dev, splitted=crystal[j - 1] and branch is None)
            b._dir_from_k = not crystal[j - 1]
            bundles.append(b)
        return _assemble_path(bundles, lengths, res)
    if split:
        return [path_of(p) for p in range(1 << sum(crystal))]
    return path_of(None)
