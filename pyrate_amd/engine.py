"""
Thin host wrapper around the C ABI: a surface table living on one GPU plus
``trace`` / ``propagate`` / ``interact`` / ``shape_eval`` / ``compact`` calls
on device-resident (3, N) float64 tensors.  torch is used only as the device
allocator and stream provider; every number is produced by ``libprt.so``.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from . import placed
from . import surface_table
from .surface_table import pack_table


COMPACT_MAX_ROWS = 24         # PRT_COMPACT_MAX_ROWS (include/prt.h)


def _torch_alloc(make):
    """a torch allocation that, out of memory, first makes the arena hand its cached buffers back to the driver
    (torch's caching allocator cannot reclaim those) and tries once more"""
    try:
        return make()
    except torch.cuda.OutOfMemoryError:
        placed.trim_all()
        torch.cuda.empty_cache()
        return make()


def _mode_word(bufs):
    return bufs["mode"] | (_lib.MODE_FLAGS if bufs.get("packed_flags") else 0)


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream_of = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def raw_stream(device):
    """address of the HIP stream that is current on ``device`` (a torch.device or an index) -- without building a
    torch.cuda.Stream object around it (1.5 us of a 30-us call)"""
    if _raw_stream_of is not None:
        index = device if isinstance(device, int) else device.index
        if index is not None:
            return _raw_stream_of(index)
    return torch.cuda.current_stream(device).cuda_stream


_current_device = getattr(torch._C, "_cuda_getDevice", torch.cuda.current_device)


def _stream_handle(device):
    return ctypes.c_void_p(raw_stream(device))


def _check_rays(t, name, n=None, allow_pitch=False):
    """(3, N) float64 device tensor with unit stride along the rays; returns the row pitch."""
    if t.dtype != torch.float64 or t.dim() != 2 or t.shape[0] != 3:
        raise ValueError("%s must be a (3, N) float64 tensor" % name)
    if not t.is_cuda:
        raise ValueError("%s must live on the GPU" % name)
    if n is not None and t.shape[1] != n:
        raise ValueError("%s has %d rays, expected %d" % (name, t.shape[1], n))
    if t.shape[1] == 0:
        return 0
    if t.stride(1) != 1 or (not allow_pitch and not t.is_contiguous()):
        raise ValueError("%s must be contiguous along the ray axis" % name)
    return t.stride(0)


def _rows_contiguous(t):
    """tight (3, N) copy of a pitched view (per-surface entry points take tight arrays)"""
    return t if (t is None or t.is_contiguous()) else t.contiguous()


def recommended_pitch(n):
    return int(_lib.load().prt_recommended_pitch(n))


class UniformFirst(object):
    """The first segment of a collimated bundle (OpticalSystemAnalysis.collimated_bundle,
    analysis/optical_system_analysis.py:83-122): ONE wave vector -- and one E field or direction -- for all
    rays.  Passed as ``uniform=`` to the whole-sequence traces instead of (3, N) arrays k0 / E0: the march then
    loads only x0 (prt_trace_ex, k0 = NULL) and produces bit-identical results.

    ``kind``: "e" -- direction = Poynting direction of (k, e); e None means E = (0,1,0) like a RayBundle
    created without a field (ray.py:71-73); "k" -- direction = k/|k|; "dir" -- ``e`` is the unit direction."""

    def __init__(self, k, e=None, kind="e"):
        if kind not in ("e", "k", "dir"):
            raise ValueError("kind must be 'e', 'k' or 'dir'")
        if kind == "dir" and e is None:
            raise ValueError("kind 'dir' needs the direction")
        self.k = tuple(float(np.real(v)) for v in k)
        e = None if e is None else np.asarray(e, dtype=complex).reshape(3)
        self.e_re = None if e is None else tuple(float(v) for v in e.real)
        self.e_im = None if e is None else tuple(float(v) for v in e.imag)
        self.kind = kind

    def _first_dir(self):
        if self.kind == "k":
            return _lib.FIRST_K
        if self.kind == "dir":
            return _lib.FIRST_DIR_UNIFORM
        return _lib.FIRST_E if self.e_re is None else _lib.FIRST_E_UNIFORM

    def fill(self, args):
        args.k0 = None
        args.e0_re = None
        args.e0_im = None
        args.first_dir = self._first_dir()
        for q in range(3):
            args.k_uniform[q] = self.k[q]
            args.e_uniform_re[q] = self.e_re[q] if self.e_re is not None else 0.0
            args.e_uniform_im[q] = self.e_im[q] if self.e_im is not None else 0.0

    def rows(self, n, device, what="k"):
        """the (3, n) array this stands for, as an expanded (stride-0) view of three numbers on the device:
        what a consumer of RayBundle.k / .Efield sees; ``.contiguous()`` materialises it"""
        v = {"k": self.k, "e_re": self.e_re, "e_im": self.e_im}[what]
        if v is None:
            return None
        return torch.tensor(v, dtype=torch.float64, device=device).view(3, 1).expand(3, n)


class _LazyViews(object):
    """list-like: per-surface tensor views created on first access (a 12-surface path has 48
    of them; an optimiser loop that only looks at the image plane should not pay for all)"""

    def __init__(self, n, make):
        self._n = n
        self._make = make
        self._cache = {}

    def __len__(self):
        return self._n

    def __getitem__(self, s):
        if isinstance(s, slice):
            return [self[i] for i in range(*s.indices(self._n))]
        if s < 0:
            s += self._n
        if not 0 <= s < self._n:
            raise IndexError(s)
        v = self._cache.get(s)
        if v is None:
            v = self._make(s)
            self._cache[s] = v
        return v

    def __iter__(self):
        return (self[i] for i in range(self._n))


class TraceResult(object):
    """Dense (uncompacted) outputs of one sequence trace.

    ``x_hit[s]`` (3, n_in[s]), ``valid[s]`` (n_in[s]) cumulative mask after
    intersect + aperture, ``k_out[s]`` (3, n_out[s]), ``valid_out[s]`` (n_out[s])
    mask carried into the next segment (what the reference compacts by).
    In image mode the lists have one entry (the last surface).
    """

    def __init__(self, x_hit, k_out, valid, valid_out, n_in, n_out, mode, e_out=None):
        self.e_out = e_out            # per surface (re, im) behind crystal interfaces (trace(want_fields))
        self.flags = None             # per surface packed mask bytes (alloc_outputs(packed_flags=True))
        self.nonconv = None           # per surface: Newton ended at its iteration cap (want_nonconv / packed flags)
        self.k_out_im = None          # crystal tables traced with want_fields: Im(k), non-zero for evanescent modes
        self.padded = None            # crystal tables: the same lists over the raw arrays with ray pitch ``ray_pitch``
        self.ray_pitch = None
        self.x_hit = x_hit
        self.k_out = k_out
        self.valid = valid
        self.valid_out = valid_out
        self.n_in = n_in
        self.n_out = n_out
        self.mode = mode

    @classmethod
    def from_buffers(cls, bufs):
        (n_in, n_out) = (bufs["n_in"], bufs["n_out"])
        rows = len(n_in)
        pitch = bufs.get("pitch", 0)
        (bx, bk, bv, bw) = (bufs["x_hit"], bufs["k_out"], bufs["valid"], bufs["valid_out"])
        if pitch and not bufs.get("concatenated"):
            n = n_in[0]
            # the last surface's record may live in redirect rows (e.g. a gather's receive buffer, _trace_args)
            img = bufs.get("image_rows") if bufs["mode"] == _lib.MODE_PATH else None
            last = rows - 1

            def rays(buf, which):
                def make(s):
                    if img is not None and s == last:
                        return img[which][:, :n]
                    return buf[3 * s * pitch:3 * (s + 1) * pitch].view(3, pitch)[:, :n]
                return make

            def mask(buf, which):
                def make(s):
                    if img is not None and s == last:
                        return img[which][:n] if len(img) > which and img[which] is not None else None
                    return buf[s * pitch:s * pitch + n]
                return make
            if bufs.get("packed_flags"):
                # one byte per record: bit 0 = valid, bit 1 = valid_out; the 0/1 masks are derived
                # on first access (and cached by _LazyViews)
                flags = mask(bv, 2)
                res = cls(_LazyViews(rows, rays(bx, 0)), _LazyViews(rows, rays(bk, 1)),
                          _LazyViews(rows, lambda s: flags(s) & 1), _LazyViews(rows, lambda s: (flags(s) >> 1) & 1),
                          n_in, n_out, bufs["mode"])
                res.flags = _LazyViews(rows, flags)
                res.nonconv = _LazyViews(rows, lambda s: (flags(s) >> 2) & 1)
                return res
            res = cls(_LazyViews(rows, rays(bx, 0)), _LazyViews(rows, rays(bk, 1)), _LazyViews(rows, mask(bv, 2)),
                      _LazyViews(rows, (mask(bw, 3) if bw is not None else (lambda s: None))),
                      n_in, n_out, bufs["mode"])
            if bufs.get("nonconv") is not None:
                bn = bufs["nonconv"]
                res.nonconv = _LazyViews(rows, lambda s: bn[s * pitch:s * pitch + n])
            return res
        # concatenated layout (tables with anisotropic media), ray pitch P >= n0: surface s holds B = n_in[s] / n0
        # branches of P slots each, the first n0 of them rays (include/prt.h).  The per-surface entries are the
        # TIGHT (3, B * n0) arrays of the reference's stacking: views when P == n0 or B == 1, otherwise gathered
        # on first access; ``padded`` gives the raw (3, B * P) arrays (what the drop-in layer compacts from).
        n0 = bufs.get("n0", n_in[0] if n_in else 0)
        P = bufs.get("pitch") or n0
        br_in = [(c // n0 if n0 else 1) for c in n_in]
        br_out = [(c // n0 if n0 else 1) for c in n_out]
        off_in = [0]
        off_out = [0]
        for (bi, bo) in zip(br_in, br_out):
            off_in.append(off_in[-1] + bi * P)
            off_out.append(off_out[-1] + bo * P)

        def rows3(buf, off, B, padded=False):
            v = buf[3 * off:3 * (off + B * P)]
            if padded:
                return v.view(3, B * P)
            if P == n0:
                return v.view(3, B * n0)
            v = v.view(3, B, P)[:, :, :n0]
            return v[:, 0, :] if B == 1 else v.reshape(3, B * n0)

        def row1(buf, off, B, padded=False):
            v = buf[off:off + B * P]
            if padded or P == n0:
                return v
            v = v.view(B, P)[:, :n0]
            return v[0] if B == 1 else v.reshape(B * n0)

        def views(padded):
            e_views = None
            if bufs.get("e_re") is not None:
                (er, ei) = (bufs["e_re"], bufs["e_im"])
                e_views = _LazyViews(rows, lambda s: (rows3(er, off_out[s], br_out[s], padded),
                                                      rows3(ei, off_out[s], br_out[s], padded)))
            res = cls(
                _LazyViews(rows, lambda s: rows3(bx, off_in[s], br_in[s], padded)),
                _LazyViews(rows, lambda s: rows3(bk, off_out[s], br_out[s], padded)),
                _LazyViews(rows, lambda s: row1(bv, off_in[s], br_in[s], padded)),
                _LazyViews(rows, (lambda s: row1(bw, off_out[s], br_out[s], padded)) if bw is not None
                           else (lambda s: None)),
                n_in, n_out, bufs["mode"], e_out=e_views)
            if bufs.get("nonconv") is not None:
                bn = bufs["nonconv"]
                res.nonconv = _LazyViews(rows, lambda s: row1(bn, off_in[s], br_in[s], padded))
            if bufs.get("k_im") is not None:
                bki = bufs["k_im"]
                res.k_out_im = _LazyViews(rows, lambda s: rows3(bki, off_out[s], br_out[s], padded))
            return res
        res = views(False)
        res.padded = views(True)
        res.ray_pitch = P
        return res


class DeviceSystem(object):
    """A flattened surface table resident on one GPU (``prt_system_t``)."""

    def __init__(self, records, device=0):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("pyrate_amd: no HIP device visible; the engine has no CPU fallback")
        self.device = torch.device("cuda", device if isinstance(device, int) else device.index or 0)
        self.records = list(records)
        self.n_surfaces = len(self.records)
        self._table = pack_table(self.records)
        # absorbing media (complex eps / complex index): complex wave vectors, per-surface march, concatenated layout,
        # TraceResult.k_out_im always there
        self.complex_eps = surface_table.has_complex_eps(self.records)
        self.all_isotropic = all(r["material"]["type"] == "isotropic" for r in self.records) and not self.complex_eps
        handle = ctypes.c_void_p()
        torch.cuda.init()
        _lib.check(self.lib.prt_system_create(self._table, self.n_surfaces,
                                              self.device.index, ctypes.byref(handle)))
        self._h = handle
        self._counts = {}          # ray_counts by n0
        self.updates = 0           # tables this system has been overwritten with (update)

    def update(self, records):
        """Replace the table IN PLACE (prt_system_update: one asynchronous copy on the current stream, no allocation).
        True if the library took the new table; False if it does not fit this system's device arrays (another
        number of surfaces, a longer coefficient array, crystals where there were none ...) -- nothing has changed
        then and the caller builds a new DeviceSystem.  Traces enqueued on the current stream before the call see
        the old table."""
        records = list(records)
        if len(records) != self.n_surfaces:
            return False
        complex_eps = surface_table.has_complex_eps(records)
        if complex_eps != self.complex_eps:
            return False
        table = pack_table(records)
        rc = self.lib.prt_system_update(self._h, table, self.n_surfaces, _stream_handle(self.device))
        if rc == _lib.ERR_UNSUPPORTED:
            return False
        if rc != 0:
            # a failed copy half way leaves the device with pieces of two tables: the library refuses the system from
            # now on, and so does this object (callers that cache systems drop it: raytracer/_dispatch.py)
            detail = self.lib.prt_last_error().decode() or self.lib.prt_strerror(rc).decode()
            self.close()
            raise _lib.PrtError(rc, detail)
        self.records = records
        self._table = table
        self._counts = {}
        self.updates += 1
        self.all_isotropic = all(r["material"]["type"] == "isotropic" for r in records) and not complex_eps
        return True

    def close(self):
        if getattr(self, "_h", None):
            self.lib.prt_system_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- bookkeeping ------------------------------------------------------
    def ray_counts(self, n0):
        hit = self._counts.get(n0)
        if hit is None:
            n_in = (ctypes.c_int64 * self.n_surfaces)()
            n_out = (ctypes.c_int64 * self.n_surfaces)()
            _lib.check(self.lib.prt_system_ray_counts(self._h, n0, n_in, n_out))
            if len(self._counts) > 64:
                self._counts.clear()
            hit = self._counts[n0] = (tuple(n_in), tuple(n_out))
        return list(hit[0]), list(hit[1])

    def alloc_outputs(self, n0, mode=_lib.MODE_PATH, with_valid_out=True, pitch=None, want_fields=False,
                      packed_flags=False, placement="auto", extra_bytes=(), want_nonconv=False, want_k_im=False):
        """Output buffers for trace_into.  All-isotropic tables get ROW-PITCHED arrays
        ((S,3,pitch) / (S,pitch), pitch = prt_recommended_pitch(n0) unless given: rows aligned
        to 128-B lines are worth ~35 % HBM write bandwidth); tables with anisotropic media get
        the concatenated layout (pitch 0).

        ``placement``: where the memory comes from.  "arena": x_hit and k_out are built from physical
        HBM slabs of two DIFFERENT kinds (``pyrate_amd.placed``; the march then writes at 7.0 instead
        of 5.6 TB/s); "torch": the torch allocator (whatever kind it happens to
        get); "auto" (default): the arena for path-mode outputs from ``placed.PLACED_MIN_BYTES`` on,
        torch otherwise.  ``extra_bytes``: further buffers to take from
        the arena in the same request (returned as uint8 tensors in ``bufs["extra"]``); the first of
        them is placed in a third kind of memory (the place for the input arrays: reads that share
        a kind with the write streams cost ~3 % of the march) -- only with arena placement."""
        if packed_flags:
            if not self.all_isotropic:
                raise ValueError("packed mask flags need an all-isotropic table")
            with_valid_out = False
        (n_in, n_out) = self.ray_counts(n0)
        if mode == _lib.MODE_IMAGE:
            n_in, n_out = n_in[-1:], n_out[-1:]
        dev = self.device
        if self.all_isotropic:
            if pitch is None:
                pitch = recommended_pitch(n0)
            rows = len(n_in)
            (nx, nk, nv, nw) = (3 * rows * pitch, 3 * rows * pitch, rows * pitch, rows * pitch)
        else:
            # concatenated layout with ray pitch (include/prt.h): n0 rounded up to 128 puts every row of every
            # level on a 128-B line (0.124 instead of 0.151 ms on BASELINE configs[3]); tight (pitch 0) for the
            # per-surface march (more crystal interfaces than the fused walk parks)
            if pitch is None:
                pitch = int(self.lib.prt_crystal_pitch(n0))
            # which march the library will run for this table is the library's decision (prt_system_layout: crystal
            # interfaces against the walk's parking slots, absorbing media, PRT_GENERAL_PER_SURFACE)
            layout = int(self.lib.prt_system_layout(self._h))
            if layout < 0:
                _lib.check(layout)
            if layout != _lib.LAYOUT_CONCATENATED_PITCHED or pitch < n0:
                pitch = 0
            P = pitch or n0
            (pin, pout) = ([c // n0 * P if n0 else 0 for c in n_in], [c // n0 * P if n0 else 0 for c in n_out])
            (nx, nk, nv, nw) = (3 * sum(pin), 3 * sum(pout), sum(pin), sum(pout))
        if not with_valid_out:
            nw = 0
        auto_placement = placement == "auto"
        if placement == "auto":
            big = 8 * (nx + nk) + nv + nw >= placed.PLACED_MIN_BYTES
            placement = "arena" if (mode == _lib.MODE_PATH and big and not want_fields
                                    and placed.DISABLED is None) else "torch"
        if placement not in ("arena", "torch"):
            raise ValueError("placement must be 'auto', 'arena' or 'torch'")
        parts = None
        if placement == "arena" and placed.DISABLED is not None:
            raise RuntimeError("placement='arena' but the arena is switched off: %s" % placed.DISABLED)
        if placement == "arena":
            # part 0: x_hit, then the mask bytes (each on a 4-KiB boundary); part 1: k_out
            def up(v):
                return -(-v // 4096) * 4096
            (off_v, off_w) = (up(8 * nx), up(up(8 * nx) + nv))
            # (without extras: stay out of the kinds beyond the first two -- that is where big input
            # bundles live, engine.ray_rows)
            try:
                arena = placed.PlacedArena.for_device(dev.index)
                (parts, kinds) = arena.alloc([off_w + nw, 8 * nk] + [int(b) for b in extra_bytes],
                                             n_distinct=3 if extra_bytes else 2,
                                             avoid_mask=0 if extra_bytes else (0xF & ~placed.OUTPUT_KINDS_MASK))
            except _lib.PrtError as exc:
                # "auto" never fails because of the arena: no whole free slabs left on the device
                # (somebody else holds the memory), or a driver without the virtual-memory API -- the
                # arrays then come from the torch allocator like small ones do (slower placement, same
                # results).  An explicit placement="arena" raises.
                if extra_bytes or not auto_placement:
                    raise
                if exc.code != _lib.ERR_NOMEM:
                    placed.disable("prt_arena_alloc failed: %s" % exc)
                placement = "torch"
        if parts is not None:
            bufs = dict(
                x_hit=parts[0][:8 * nx].view(torch.float64),
                k_out=parts[1][:8 * nk].view(torch.float64),
                valid=parts[0][off_v:off_v + nv],
                valid_out=(parts[0][off_w:off_w + nw] if with_valid_out else None),
                extra=parts[2:], placement={"policy": "arena", "kinds": kinds})
        else:
            if extra_bytes:
                raise ValueError("extra_bytes needs arena placement")
            bufs = _torch_alloc(lambda: dict(
                x_hit=torch.empty(nx, dtype=torch.float64, device=dev),
                k_out=torch.empty(nk, dtype=torch.float64, device=dev),
                valid=torch.empty(nv, dtype=torch.uint8, device=dev),
                valid_out=(torch.empty(nw, dtype=torch.uint8, device=dev) if with_valid_out else None),
                extra=[], placement={"policy": "torch"}))
        bufs.update(n_in=n_in, n_out=n_out, mode=mode, pitch=pitch, packed_flags=bool(packed_flags), n0=n0,
                    concatenated=not self.all_isotropic)
        if not self.all_isotropic and pitch and pitch != n0:
            # the padding slots of a branch are never written by the engine: their masks say "no ray"
            bufs["valid"].zero_()
            if bufs["valid_out"] is not None:
                bufs["valid_out"].zero_()
        if want_nonconv and not packed_flags:      # packed flags carry the bit themselves (bit 2)
            bufs["nonconv"] = torch.zeros(nv, dtype=torch.uint8, device=dev)
        if want_fields:
            if self.all_isotropic:
                raise ValueError("E fields are produced at crystal interfaces only")
            bufs["e_re"] = torch.zeros(nk, dtype=torch.float64, device=dev)
            bufs["e_im"] = torch.zeros(nk, dtype=torch.float64, device=dev)
            if mode == _lib.MODE_PATH and pitch:      # (the fused march; the per-surface march has no such report)
                # imaginary parts of the wave vectors: non-zero in the slots of evanescent modes (prt.h k_out_im;
                # the library zeroes the array itself before its post-pass fills those slots)
                bufs["k_im"] = torch.empty(nk, dtype=torch.float64, device=dev)
        if want_k_im and not want_fields and not self.all_isotropic and mode == _lib.MODE_PATH and pitch:
            # the complex wave vectors of evanescent modes alone (the fused march's post-pass), without the E fields
            bufs["k_im"] = torch.empty(nk, dtype=torch.float64, device=dev)
        if self.complex_eps:
            # absorbing crystals: the wave vectors ARE complex (prt.h: k_out_im is required for such tables)
            bufs["k_im"] = torch.zeros(nk, dtype=torch.float64, device=dev)
        return bufs

    # -- whole sequence ----------------------------------------------------
    def _trace_args(self, x0, k0, bufs, e0_re=None, e0_im=None, uniform=None, first_dir=None):
        """prt_trace_args_t for one trace into ``bufs``.  ``uniform`` (a UniformFirst) replaces k0 / E0.
        ``bufs["image_rows"]`` = (x_img, k_img, mask_img) -- (3, n) / (n,) views with unit stride along the rays,
        e.g. ``ImagePlaneGather.own_rows()`` -- redirects the last surface's record there (path mode)."""
        n0 = x0.shape[1]
        if uniform is None and k0 is None:
            raise ValueError("k0 is required (or a uniform first segment)")
        if uniform is not None:
            (k0, e0_re, e0_im) = (None, None, None)
        if not self.all_isotropic:
            # the fused crystal march takes row-pitched inputs; the per-surface march (more crystal interfaces than
            # the walk parks) and mixed pitches need tight arrays
            strides = set(t.stride(0) for t in (x0, k0, e0_re, e0_im) if t is not None and t.shape[1] > 0)
            if len(strides) > 1 or not bufs.get("pitch") or any(t is not None and t.shape[1] > 0 and t.stride(1) != 1
                                                                 for t in (x0, k0, e0_re, e0_im)):
                (x0, k0, e0_re, e0_im) = [_rows_contiguous(t) for t in (x0, k0, e0_re, e0_im)]
        a = _lib.PrtTraceArgs()
        a.struct_bytes = ctypes.sizeof(_lib.PrtTraceArgs)
        a.mode = _mode_word(bufs)
        a.n0 = n0
        a.in_pitch = self._in_pitch(x0, k0, e0_re, e0_im)
        a.x0 = x0.data_ptr() if n0 else None
        if uniform is not None:
            uniform.fill(a)
        else:
            a.k0 = k0.data_ptr() if n0 else None
            a.e0_re = None if e0_re is None else e0_re.data_ptr()
            a.e0_im = None if e0_im is None else e0_im.data_ptr()
            a.first_dir = _lib.FIRST_E if first_dir is None else first_dir
        a.out_pitch = bufs["pitch"]
        for name in ("x_hit", "k_out", "valid", "valid_out", "nonconv"):
            t = bufs.get(name)
            setattr(a, name, None if t is None else t.data_ptr())
        if bufs.get("e_re") is not None:
            a.e_out_re = bufs["e_re"].data_ptr()
            # Im(E) is zero for a real epsilon and real wave vectors: the array (zeros from alloc_outputs) is handed to
            # the march only for tables with absorbing media -- 24 B per leaving ray the lossless march need not write
            a.e_out_im = bufs["e_im"].data_ptr() if self.complex_eps else None
        if bufs.get("k_im") is not None:
            a.k_out_im = bufs["k_im"].data_ptr()
        img = bufs.get("image_rows")
        if img is not None:
            (xi, ki, vi) = img[:3]
            if xi.stride(0) != ki.stride(0) or (n0 and (xi.stride(1) != 1 or ki.stride(1) != 1 or vi.stride(0) != 1)):
                raise ValueError("image rows: x and k must share one row pitch and have unit stride along the rays")
            a.x_img = xi.data_ptr()
            a.k_img = ki.data_ptr()
            a.valid_img = vi.data_ptr()
            a.valid_out_img = img[3].data_ptr() if len(img) > 3 and img[3] is not None else None
            a.img_pitch = xi.stride(0)
        a.stream = raw_stream(self.device)
        a._keep = (x0, k0, e0_re, e0_im)       # tight copies must outlive the launch call
        return a

    def trace_into(self, x0, k0, bufs, e0_re=None, e0_im=None, uniform=None, first_dir=None):
        """Asynchronous launch into preallocated buffers (see alloc_outputs).  ``uniform``: a UniformFirst
        instead of the arrays k0 / e0 (collimated bundles); ``first_dir``: _lib.FIRST_K / FIRST_DIR for
        bundles whose first direction is k/|k| / given in e0_re."""
        a = self._trace_args(x0, k0, bufs, e0_re, e0_im, uniform, first_dir)
        _lib.check(self.lib.prt_trace_ex(self._h, ctypes.byref(a)))

    def launcher(self, x0, k0, bufs, e0_re=None, e0_im=None, uniform=None, first_dir=None):
        """a callable that enqueues this trace on the current stream: the argument struct is built once, a call
        costs one ctypes call (repeated traces into the same arrays: timing loops, wavelength sweeps; the
        march of a small crystal bundle takes 0.12 ms, building the struct in Python about as long)"""
        a = self._trace_args(x0, k0, bufs, e0_re, e0_im, uniform, first_dir)
        (fn, h, ref, dev) = (self.lib.prt_trace_ex, self._h, ctypes.byref(a), self.device)

        def launch():
            a.stream = raw_stream(dev)
            rc = fn(h, ref)
            if rc < 0:
                _lib.check(rc)
        launch.args = a
        return launch

    def trace_moments_into(self, x0, k0, bufs, ws, slot=0, e0_re=None, e0_im=None, ref=None, uniform=None,
                           first_dir=None):
        """trace_into + the image-plane moments of the traced bundle from the same launch
        (prt_trace_moments): ws.out[slot] = {count, sum v, sum v*v}, v = x_img - ref (default: vertex
        of the last surface), over the rays valid after the last surface.  See ``spot_from_moments``."""
        n0 = x0.shape[1]
        need = self.lib.prt_trace_moments_scratch_doubles(n0)
        if ws.scratch.numel() < need:
            raise ValueError("MomentsWorkspace too small: construct it with n_rays >= %d" % n0)
        a = self._trace_args(x0, k0, bufs, e0_re, e0_im, uniform, first_dir)
        ref3 = None if ref is None else (ctypes.c_double * 3)(*[float(v) for v in ref])
        if ref3 is not None:
            a.moments_ref3 = ctypes.cast(ref3, ctypes.POINTER(ctypes.c_double))
        a.moments_out7_dev = ws.out[slot].data_ptr()
        a.moments_scratch_dev = ws.scratch.data_ptr()
        _lib.check(self.lib.prt_trace_ex(self._h, ctypes.byref(a)))
        return ws.out[slot]

    def moments_reference(self):
        """default reference point of trace_moments_into: global vertex of the last surface"""
        return [float(v) for v in self.records[-1]["g_shape"]]

    @staticmethod
    def _in_pitch(x0, k0, e0_re, e0_im):
        if x0.shape[1] == 0:
            return 0
        pitch = x0.stride(0)
        for t in (k0, e0_re, e0_im):
            if t is not None and t.stride(0) != pitch:
                raise ValueError("x0, k0 and E0 must share one row pitch")
        return pitch

    def trace_timed(self, x0, k0, bufs, iters, e0_re=None, e0_im=None, uniform=None):
        """Average device milliseconds per trace launch (HIP events on the launch stream, inside libprt)."""
        ms = ctypes.c_double()
        a = self._trace_args(x0, k0, bufs, e0_re, e0_im, uniform)
        a.timed_iters = iters
        a.ms_avg = ctypes.pointer(ms)
        _lib.check(self.lib.prt_trace_ex(self._h, ctypes.byref(a)))
        return ms.value

    def trace(self, x0, k0, e0_re=None, e0_im=None, mode=_lib.MODE_PATH, want_fields=False,
              packed_flags=False, want_nonconv=False, uniform=None, first_dir=None, want_k_im=False):
        """OpticalSystem.seqtrace on device tensors; returns a TraceResult of views.
        ``uniform``: a UniformFirst instead of k0 / e0 (collimated bundle: k0 may be None).
        ``want_k_im`` (tables with crystals, without ``want_fields``): ``TraceResult.k_out_im``, the imaginary parts
        of the wave vectors of evanescent modes, WITHOUT the E fields -- the march then computes no eigenvectors.
        ``want_nonconv``: also fill ``TraceResult.nonconv`` (per surface, 1 where the Newton iteration of
        an explicit shape ended at its cap; with ``packed_flags`` it is always there, bit 2 of the flags)."""
        n0 = x0.shape[1]
        if uniform is not None:
            (k0, e0_re, e0_im) = (None, None, None)
        pitches = set()
        for (t, name) in ((x0, "x0"), (k0, "k0"), (e0_re, "e0_re"), (e0_im, "e0_im")):
            if t is not None:
                pitches.add(_check_rays(t, name, n0, allow_pitch=True))
        if len(pitches) > 1:
            # mixed pitches: tight copies (the per-surface march through many crystals gets them in _trace_args)
            (x0, k0, e0_re, e0_im) = [_rows_contiguous(t) for t in (x0, k0, e0_re, e0_im)]
        if _current_device() == self.device.index:        # (the context manager costs 2 us of a 30-us call)
            bufs = self.alloc_outputs(n0, mode, want_fields=want_fields,
                                      packed_flags=packed_flags and self.all_isotropic,
                                      want_nonconv=want_nonconv and not want_fields, want_k_im=want_k_im)
            self.trace_into(x0, k0, bufs, e0_re, e0_im, uniform=uniform, first_dir=first_dir)
            return self.views(bufs)
        with torch.cuda.device(self.device):
            bufs = self.alloc_outputs(n0, mode, want_fields=want_fields,
                                      packed_flags=packed_flags and self.all_isotropic,
                                      want_nonconv=want_nonconv and not want_fields, want_k_im=want_k_im)
            self.trace_into(x0, k0, bufs, e0_re, e0_im, uniform=uniform, first_dir=first_dir)
        return self.views(bufs)

    @staticmethod
    def views(bufs):
        return TraceResult.from_buffers(bufs)

    # -- per-surface plugin granularity -------------------------------------
    def _step_arrays(self, n, reads, placement, n_masks=1):
        """output arrays of one per-surface call on ``n`` rays: a (3, n) float64 array and ``n_masks`` byte rows.
        Big bundles (``placement`` "auto": from placed.PLACED_INPUT_MIN_BYTES per array on) get ROW-PITCHED arrays
        (rows on 128-B lines) out of arena memory of a kind that none of the arrays in ``reads`` lies in
        (placed.RowPool); everything else comes from the torch allocator, tight."""
        dev = self.device
        big = 24 * n >= placed.PLACED_INPUT_MIN_BYTES
        if placement not in ("auto", "arena", "torch"):
            raise ValueError("placement must be 'auto', 'arena' or 'torch'")
        if placement == "arena" and placed.DISABLED is not None:
            raise RuntimeError("placement='arena' but the arena is switched off: %s" % placed.DISABLED)
        if placement == "arena" or (placement == "auto" and big and placed.DISABLED is None):
            try:
                arena = placed.PlacedArena.for_device(dev.index)
                avoid = [arena.kind_of(t) for t in reads if t is not None]
                pitch = recommended_pitch(n)
                mrow = -(-n // 4096) * 4096
                (buf, _) = placed.row_pool.take(dev, 24 * pitch + n_masks * mrow, avoid)
                arr = buf[:24 * pitch].view(torch.float64).view(3, pitch)[:, :n]
                masks = [buf[24 * pitch + q * mrow:24 * pitch + q * mrow + n] for q in range(n_masks)]
                return arr, masks
            except _lib.PrtError as exc:
                if placement == "arena":
                    raise
                if exc.code != _lib.ERR_NOMEM:
                    placed.disable("prt_arena_alloc failed: %s" % exc)
        return _torch_alloc(lambda: (torch.empty((3, n), dtype=torch.float64, device=dev),
                                     [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(n_masks)]))

    @staticmethod
    def _row_pitch(t, name):
        """row pitch of a (3, n) array that the rows entry points can take as it is; None: needs a tight copy"""
        if t is None or t.shape[1] == 0:
            return 0
        if t.dtype != torch.float64 or t.dim() != 2 or t.shape[0] != 3 or not t.is_cuda:
            raise ValueError("%s must be a (3, N) float64 device tensor" % name)
        return t.stride(0) if (t.stride(1) == 1 and t.stride(0) >= t.shape[1]) else None

    def propagate(self, surface, x, k, direction=None, e_re=None, e_im=None,
                  default_e=True, valid_in=None, want_nonconv=False, placement="auto"):
        """Material.propagate / Surface.intersect for one surface.  Returns (x_hit, valid), with
        ``want_nonconv`` (x_hit, valid, nonconv).  (3, n) arrays with a row pitch are taken as they are.
        ``placement``: where x_hit comes from -- "auto": big bundles get a row-pitched array in arena memory of another
        kind of HBM than the arrays the call reads (``_step_arrays``); "torch": tight, from the torch allocator."""
        if self._row_pitch(x, "x") is None:
            x = x.contiguous()
        # k, the direction and E share one pitch in the entry point
        group = [t for t in (k, direction, e_re, e_im) if t is not None]
        pitches = set(self._row_pitch(t, "k / direction / E") for t in group)
        if None in pitches or len(pitches) > 1:
            (k, direction, e_re, e_im) = [_rows_contiguous(t) for t in (k, direction, e_re, e_im)]
        n = x.shape[1]
        lead = direction if direction is not None else k
        with torch.cuda.device(self.device):
            (x_hit, masks) = self._step_arrays(n, (x, lead), placement, n_masks=2 if want_nonconv else 1)
            valid = masks[0]
            nonconv = masks[1] if want_nonconv else None
            _lib.check(self.lib.prt_propagate_rows(self._h, surface, n, _ptr(x), x.stride(0) if n else 0, _ptr(k),
                                                   lead.stride(0) if (n and lead is not None) else 0,
                                                   _ptr(direction), _ptr(e_re), _ptr(e_im),
                                                   1 if default_e else 0, _ptr(valid_in), _ptr(x_hit),
                                                   x_hit.stride(0) if n else 0, _ptr(valid), _ptr(nonconv),
                                                   _stream_handle(self.device)))
        return (x_hit, valid, nonconv) if want_nonconv else (x_hit, valid)

    def surface_step(self, surface, x, k, direction=None, e_re=None, e_im=None, default_e=True, valid_in=None,
                     want_nonconv=False, placement="auto"):
        """``propagate`` + ``interact`` of ONE surface in one launch (prt_surface_step_rows): the loop body of
        OpticalElement.seqtrace (optical_element.py:336-375) for a surface with an isotropic, lossless medium behind
        it.  Returns (x_hit, k_out, valid, valid_out), with ``want_nonconv`` also nonconv.  98 B of HBM traffic per
        ray (the 49-B state read once, the 49-B record written once) against the 148 B of the two separate calls.
        ``placement`` "auto": big bundles get row-pitched arrays from the arena, x_hit in the kind of HBM that neither
        x nor k lies in, k_out in x's kind -- the two write streams in two kinds, like the fused march."""
        if self.records[surface]["material"]["type"] == "anisotropic" or self.complex_eps:
            raise ValueError("surface_step: isotropic, lossless media only (crystals: propagate + interact)")
        if self._row_pitch(x, "x") is None:
            x = x.contiguous()
        group = [t for t in (k, direction, e_re, e_im) if t is not None]
        pitches = set(self._row_pitch(t, "k / direction / E") for t in group)
        if None in pitches or len(pitches) > 1:
            (k, direction, e_re, e_im) = [_rows_contiguous(t) for t in (k, direction, e_re, e_im)]
        n = x.shape[1]
        with torch.cuda.device(self.device):
            (x_hit, m1) = self._step_arrays(n, (x, k), placement, n_masks=2 if want_nonconv else 1)
            (k_out, m2) = self._step_arrays(n, (k, x_hit), placement, n_masks=1)
            if n and k_out.stride(0) != x_hit.stride(0):          # (one pitch for both outputs in the entry point)
                k_out = torch.empty((3, x_hit.stride(0)), dtype=torch.float64, device=self.device)[:, :n]
            nonconv = m1[1] if want_nonconv else None
            _lib.check(self.lib.prt_surface_step_rows(self._h, surface, n, _ptr(x), x.stride(0) if n else 0, _ptr(k),
                                                      k.stride(0) if n else 0, _ptr(direction), _ptr(e_re), _ptr(e_im),
                                                      1 if default_e else 0, _ptr(valid_in), _ptr(x_hit), _ptr(k_out),
                                                      x_hit.stride(0) if n else 0, _ptr(m1[0]), _ptr(m2[0]),
                                                      _ptr(nonconv), _stream_handle(self.device)))
        return (x_hit, k_out, m1[0], m2[0], nonconv) if want_nonconv else (x_hit, k_out, m1[0], m2[0])

    def interact(self, surface, x_hit, k, valid_in=None, want_e=False, want_dir=None, placement="auto"):
        """Material.refract / reflect at one surface.  Returns (k_out, dir_out, valid_out, e_re, e_im).
        Isotropic medium behind the surface: ``dir_out`` is None unless ``want_dir`` (the ray direction is k / |k| there,
        and the next ``propagate`` takes k itself); row-pitched arrays are taken as they are; ``placement`` as in
        ``propagate``.  Crystals: (3, 2n) tight outputs, ``dir_out`` always."""
        aniso = self.records[surface]["material"]["type"] == "anisotropic"
        n = x_hit.shape[1]
        if aniso or self.complex_eps:
            (x_hit, k) = [_rows_contiguous(t) for t in (x_hit, k)]
            _check_rays(x_hit, "x_hit")
            m = 2 * n if aniso else n
            with torch.cuda.device(self.device):
                k_out = torch.empty((3, m), dtype=torch.float64, device=self.device)
                dir_out = torch.empty((3, m), dtype=torch.float64, device=self.device)
                valid_out = torch.empty(m, dtype=torch.uint8, device=self.device)
                e_re = e_im = None
                if aniso and want_e:
                    e_re = torch.empty((3, m), dtype=torch.float64, device=self.device)
                    e_im = torch.empty((3, m), dtype=torch.float64, device=self.device)
                _lib.check(self.lib.prt_interact(self._h, surface, n, _ptr(x_hit), _ptr(k),
                                                 _ptr(valid_in), _ptr(k_out), _ptr(dir_out),
                                                 _ptr(e_re), _ptr(e_im), _ptr(valid_out),
                                                 _stream_handle(self.device)))
            return k_out, dir_out, valid_out, e_re, e_im
        if self._row_pitch(x_hit, "x_hit") is None:
            x_hit = x_hit.contiguous()
        if self._row_pitch(k, "k") is None:
            k = k.contiguous()
        with torch.cuda.device(self.device):
            (k_out, masks) = self._step_arrays(n, (x_hit, k), placement)
            dir_out = None
            if want_dir:
                (dir_out, _) = self._step_arrays(n, (x_hit, k), "torch", n_masks=0)
                if k_out.stride(0) != dir_out.stride(0):        # (dir_out shares k_out's pitch in the entry point)
                    dir_out = torch.empty((3, k_out.stride(0)), dtype=torch.float64, device=self.device)[:, :n]
            _lib.check(self.lib.prt_interact_rows(self._h, surface, n, _ptr(x_hit), x_hit.stride(0) if n else 0, _ptr(k),
                                                  k.stride(0) if n else 0, _ptr(valid_in), _ptr(k_out),
                                                  k_out.stride(0) if n else 0, _ptr(dir_out), _ptr(masks[0]),
                                                  _stream_handle(self.device)))
        return k_out, dir_out, masks[0], None, None

    def interact_cplx(self, surface, x_hit, k_re, k_im=None, valid_in=None, want_e=False):
        """Material.refract / reflect at one surface with complex wave vectors (prt_interact_cplx: absorbing media
        and what comes behind them).  Returns (k_out_re, k_out_im, dir_out, valid_out, e_re, e_im); ``dir_out`` is
        None behind an isotropic interface (include/prt.h)."""
        (x_hit, k_re, k_im) = [_rows_contiguous(t) for t in (x_hit, k_re, k_im)]
        _check_rays(x_hit, "x_hit")
        n = x_hit.shape[1]
        aniso = self.records[surface]["material"]["type"] == "anisotropic"
        m = 2 * n if aniso else n
        with torch.cuda.device(self.device):
            k_out = torch.empty((3, m), dtype=torch.float64, device=self.device)
            k_out_im = torch.empty((3, m), dtype=torch.float64, device=self.device)
            dir_out = torch.empty((3, m), dtype=torch.float64, device=self.device) if aniso else None
            valid_out = torch.empty(m, dtype=torch.uint8, device=self.device)
            e_re = e_im = None
            if aniso and want_e:
                e_re = torch.empty((3, m), dtype=torch.float64, device=self.device)
                e_im = torch.empty((3, m), dtype=torch.float64, device=self.device)
            _lib.check(self.lib.prt_interact_cplx(self._h, surface, n, _ptr(x_hit), _ptr(k_re), _ptr(k_im),
                                                  _ptr(valid_in), _ptr(k_out), _ptr(k_out_im), _ptr(dir_out),
                                                  _ptr(e_re), _ptr(e_im), _ptr(valid_out),
                                                  _stream_handle(self.device)))
        return k_out, k_out_im, dir_out, valid_out, e_re, e_im

    def shape_eval(self, surface, x, y, want_sag=True, want_grad=True):
        """Shape.getSag / getGrad on the device; x, y 1-d float64 tensors (shape frame)."""
        n = x.shape[0]
        with torch.cuda.device(self.device):
            sag = torch.empty(n, dtype=torch.float64, device=self.device) if want_sag else None
            grad = torch.empty((3, n), dtype=torch.float64, device=self.device) if want_grad else None
            _lib.check(self.lib.prt_shape_eval(self._h, surface, n, _ptr(x), _ptr(y), _ptr(sag),
                                               _ptr(grad), _stream_handle(self.device)))
        return sag, grad


def trace_seq(records, x0, k0, d0=None, mode=_lib.MODE_PATH, want_nonconv=False, device=None):
    """The one-call form of the boundary (prt_trace_seq): surface records + tight (3, n)
    device arrays in, dense tight outputs back -- no DeviceSystem to keep; the library caches the uploaded
    table by content.  ``d0``: unit directions of the first segment, None = k/|k|.  Returns a TraceResult
    (x_hit, k_out, valid [, nonconv]; valid_out is not part of this entry point)."""
    lib = _lib.load()
    (x0, k0, d0) = [_rows_contiguous(t) for t in (x0, k0, d0)]
    _check_rays(x0, "x0")
    n = x0.shape[1]
    dev = x0.device if device is None else device
    table = pack_table(list(records))
    S = len(table)
    counts_in = [n]
    for r in list(records)[:-1]:
        counts_in.append(counts_in[-1] * (2 if r["material"]["type"] == "anisotropic" else 1))
    counts_out = [c * (2 if r["material"]["type"] == "anisotropic" else 1) for (c, r) in zip(counts_in, records)]
    if mode == _lib.MODE_IMAGE:
        (counts_in, counts_out) = (counts_in[-1:], counts_out[-1:])
    with torch.cuda.device(dev):
        bufs = dict(x_hit=torch.empty(3 * sum(counts_in), dtype=torch.float64, device=dev),
                    k_out=torch.empty(3 * sum(counts_out), dtype=torch.float64, device=dev),
                    valid=torch.empty(sum(counts_in), dtype=torch.uint8, device=dev), valid_out=None,
                    n_in=counts_in, n_out=counts_out, mode=mode, pitch=0, packed_flags=False, n0=n,
                    concatenated=True)
        if want_nonconv:
            bufs["nonconv"] = torch.zeros(sum(counts_in), dtype=torch.uint8, device=dev)
        _lib.check(lib.prt_trace_seq(table, S, n, _ptr(x0), _ptr(k0), _ptr(d0), None, mode, _ptr(bufs["x_hit"]),
                                     _ptr(bufs["k_out"]), _ptr(bufs["valid"]), _ptr(bufs.get("nonconv")),
                                     dev.index, _stream_handle(dev)))
    return TraceResult.from_buffers(bufs)


def compact(mask, arrays, ids=None, flags=None):
    """Order-preserving ``[:, mask]`` on the device (material_isotropic.py:194-199).
    arrays: list of (R_i, N) float64 tensors; ids: optional (N,) int64; flags: optional
    (N,) uint8.  Returns (list of compacted tensors, compacted ids or None, compacted flags
    or None); the tensors are row-pitched views of one allocation."""
    lib = _lib.load()
    n = mask.shape[0]
    dev = mask.device
    rows_src = []
    for a in arrays:
        if a.dtype != torch.float64 or a.dim() != 2 or a.shape[1] != n or (n and a.stride(1) != 1):
            raise ValueError("compact: arrays must be float64 (R, N) with unit stride along N")
        rows_src += [a[r] for r in range(a.shape[0])]
    nrow = len(rows_src)
    if nrow > COMPACT_MAX_ROWS:
        raise ValueError("compact: at most %d rows per call" % COMPACT_MAX_ROWS)
    with torch.cuda.device(dev):
        tmp = _torch_alloc(lambda: torch.empty((max(nrow, 1), n), dtype=torch.float64, device=dev))
        idt = torch.empty(n, dtype=torch.int64, device=dev) if ids is not None else None
        flt = torch.empty(n, dtype=torch.uint8, device=dev) if flags is not None else None
        scratch = torch.empty(lib.prt_compact_scratch_bytes(n), dtype=torch.uint8, device=dev)
        src = (ctypes.c_void_p * max(nrow, 1))(*[r.data_ptr() for r in rows_src])
        dst = (ctypes.c_void_p * max(nrow, 1))(*[tmp[r].data_ptr() for r in range(nrow)])
        kept = ctypes.c_int64()
        _lib.check(lib.prt_compact(n, _ptr(mask), nrow, src, dst, _ptr(ids), _ptr(idt),
                                   _ptr(flags), _ptr(flt), _ptr(scratch), ctypes.byref(kept),
                                   _stream_handle(dev)))
    m = kept.value
    if 2 * m < n and nrow:
        # heavily vignetted bundle: a right-sized copy instead of views that pin rows * N * 8 bytes
        # for as long as the compacted bundle lives
        small = torch.empty((nrow, recommended_pitch(max(m, 1))), dtype=torch.float64, device=dev)
        small[:, :m].copy_(tmp[:, :m])
        tmp = small
    out = []
    r0 = 0
    for a in arrays:
        rr = a.shape[0]
        out.append(tmp[r0:r0 + rr, :m])        # row-pitched view, no second copy when most rays survive
        r0 += rr
    idc = idt[:m].contiguous() if ids is not None else None
    flc = flt[:m].contiguous() if flags is not None else None
    return out, idc, flc


def bundle_moments(x, mask=None, ref=None, mode=0):
    """(count, sum(v) (3,), sum(v**2) (3,)) of a (3, N) device array over the rays with
    mask != 0 (prt_bundle_moments); mode 0: v = x - ref, 1: v = x/|x|, 2: v = x/|x| cross ref.
    x may be a row-pitched view."""
    lib = _lib.load()
    pitch = _check_rays(x, "x", allow_pitch=True)
    n = x.shape[1]
    out = (ctypes.c_double * 7)()
    refc = None
    if ref is not None:
        refc = (ctypes.c_double * 3)(*[float(v) for v in ref])
    if mask is not None and (mask.dtype != torch.uint8 or not mask.is_contiguous() or mask.shape[0] != n):
        raise ValueError("mask must be a contiguous (N,) uint8 tensor")
    with torch.cuda.device(x.device):
        _lib.check(lib.prt_bundle_moments(x.device.index, n, pitch, _ptr(x), _ptr(mask), mode, refc, out,
                                          _stream_handle(x.device)))
    v = np.array(list(out))
    return v[0], v[1:4], v[4:7]


def rect_grid_count(nray, device):
    """(samples per dimension, points inside the unit disk) of RectGrid.getGrid(nray)"""
    lib = _lib.load()
    n_per_dim = ctypes.c_int64()
    n_disk = ctypes.c_int64()
    with torch.cuda.device(device):
        _lib.check(lib.prt_rect_grid_count(device.index, int(nray), ctypes.byref(n_per_dim),
                                           ctypes.byref(n_disk), _stream_handle(device)))
    return n_per_dim.value, n_disk.value


def collimated_bundle_device(nray, radius, start, kvec, evec, device, lo=0, hi=None, uniform=False):
    """RectGrid raster + collimated bundle generated on the GPU (prt_collimated_bundle).
    Returns row-pitched (3, hi-lo) views x, k, e and the total number of rays in the raster; with
    ``uniform`` only x is generated and (x, UniformFirst(kvec, evec), None, total) is returned."""
    lib = _lib.load()
    (_, total) = rect_grid_count(nray, device)
    if hi is None:
        hi = total
    n = hi - lo
    prm = _lib.PrtCollimated()
    (prm.radius, prm.startx, prm.starty, prm.startz) = (float(radius), float(start[0]), float(start[1]),
                                                       float(start[2]))
    for q in range(3):
        prm.k[q] = float(kvec[q])
        prm.e[q] = float(evec[q])
    pitch = recommended_pitch(n)
    with torch.cuda.device(device):
        bufs = [ray_rows(max(n, 1), device) for _ in range(1 if uniform else 3)] + [None, None]
        _lib.check(lib.prt_collimated_bundle(device.index, int(nray), lo, hi, ctypes.byref(prm), pitch,
                                             _ptr(bufs[0]), _ptr(bufs[1]), _ptr(bufs[2]),
                                             _stream_handle(device)))
    if uniform:
        return bufs[0][:, :n], UniformFirst(kvec, evec, "e"), None, total
    return bufs[0][:, :n], bufs[1][:, :n], bufs[2][:, :n], total


def _raster_struct(tables):
    """prt_raster_t for one (xa, xb, ya, yb, clip) entry of ``raster.device_tables`` (keeps the arrays alive)"""
    (xa, xb, ya, yb, clip) = tables
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (xa, xb, ya, yb)]
    r = _lib.PrtRaster()
    (r.nj, r.ni) = (arrs[0].shape[0], arrs[1].shape[0])
    if arrs[2].shape[0] != r.nj or arrs[3].shape[0] != r.ni:
        raise ValueError("raster tables: xa, ya need nj entries, xb, yb ni entries")
    (r.xa, r.xb, r.ya, r.yb) = [a.ctypes.data for a in arrs]
    r.clip = 1 if clip else 0
    r._keepalive = arrs
    return r


def raster_bundle_device(tables_list, kind, device, radius=1.0, start=(0., 0., 0.), anglex=0.0, angley=0.0,
                         index=1.0, kvec=None, evec=None, lo=0, hi=None, want_pupil=False, uniform=False):
    """A bundle on a pupil raster given by its outer-product tables (``raster.device_tables(nray)``),
    generated on the GPU (prt_raster_bundle): ``kind`` "collimated" (origin = radius * p + start, kvec /
    evec constant) or "divergent" (origin = start, directions fanned out over the pupil angles, k = index *
    unit vector).  Rays [lo, hi) of the concatenated sub-rasters.  Returns row-pitched (3, n) views
    (x, k, e), the total number of points and -- with ``want_pupil`` -- the (2, n) pupil samples.  With
    ``uniform`` (collimated only) k and e are not generated: (x, UniformFirst(kvec, evec), None, total ...)."""
    lib = _lib.load()
    rasters = [_raster_struct(t) for t in tables_list]
    counts = []
    with torch.cuda.device(device):
        for r in rasters:
            c = ctypes.c_int64()
            _lib.check(lib.prt_raster_count(device.index, ctypes.byref(r), ctypes.byref(c), _stream_handle(device)))
            counts.append(c.value)
    total = sum(counts)
    if hi is None:
        hi = total
    if not 0 <= lo <= hi <= total:
        raise ValueError("ray range [%d, %d) outside the raster's %d points" % (lo, hi, total))
    n = hi - lo
    prm = _lib.PrtBundle()
    prm.kind = {"collimated": 0, "divergent": 1}[kind]
    (prm.radius, prm.anglex, prm.angley, prm.index) = (float(radius), float(anglex), float(angley), float(index))
    for q in range(3):
        prm.start[q] = float(start[q])
        prm.k[q] = float(kvec[q]) if kvec is not None else 0.0
        prm.e[q] = float(evec[q]) if evec is not None else 0.0
    pitch = recommended_pitch(max(n, 1))
    if uniform and kind != "collimated":
        raise ValueError("only collimated bundles have a uniform first segment")
    bufs = [ray_rows(max(n, 1), device) for _ in range(1 if uniform else 3)]
    pup = torch.empty((2, pitch), dtype=torch.float64, device=device) if want_pupil else None
    with torch.cuda.device(device):
        base = 0                   # global index of the current sub-raster's first point
        for (r, c) in zip(rasters, counts):
            (a, b) = (max(lo, base), min(hi, base + c))
            if a < b:
                off = (a - lo) * 8
                _lib.check(lib.prt_raster_bundle(
                    device.index, ctypes.byref(r), a - base, b - base, ctypes.byref(prm), pitch,
                    ctypes.c_void_p(bufs[0].data_ptr() + off),
                    None if uniform else ctypes.c_void_p(bufs[1].data_ptr() + off),
                    None if uniform else ctypes.c_void_p(bufs[2].data_ptr() + off),
                    None if pup is None else ctypes.c_void_p(pup.data_ptr() + off), _stream_handle(device)))
            base += c
    if uniform:
        out = (bufs[0][:, :n], UniformFirst(kvec, evec, "e"), None, total)
    else:
        out = (bufs[0][:, :n], bufs[1][:, :n], bufs[2][:, :n], total)
    return out + (pup[:, :n],) if want_pupil else out


class MomentsWorkspace(object):
    """device scratch + result vectors for bundle_moments_async (no allocation per call)"""

    def __init__(self, device, n_results=2, n_rays=0, host_results=False):
        """n_rays: size the scratch for DeviceSystem.trace_moments_into of bundles up to n_rays.
        ``host_results``: the result vectors live in page-locked HOST memory that the device writes directly (56
        bytes across PCIe from the last reduction kernel; the caller synchronises the stream and reads
        ``host[slot]``, a NumPy view) -- for results that go to the host anyway (a merit function) this saves the
        device-to-host copy call; results that feed a collective stay on the device (default)."""
        lib = _lib.load()
        self.device = device
        self.scratch = torch.empty(max(lib.prt_moments_scratch_doubles(0),
                                       lib.prt_trace_moments_scratch_doubles(n_rays)),
                                   dtype=torch.float64, device=device)
        self.host = None
        if host_results:
            self.out = [torch.zeros(7, dtype=torch.float64, pin_memory=True) for _ in range(n_results)]
            self.host = [t.numpy() for t in self.out]
        else:
            self.out = [torch.zeros(7, dtype=torch.float64, device=device) for _ in range(n_results)]


def bundle_moments_async(x, mask, ws, slot=0, mode=0, ref_dev=None, ref_kind=0):
    """asynchronous prt_bundle_moments on the current stream; the 7-vector stays on the device
    (ws.out[slot]).  ref_kind 1: ref_dev = reference point (3,), 2: ref_dev = a moments vector whose
    centroid is the reference."""
    lib = _lib.load()
    pitch = _check_rays(x, "x", allow_pitch=True)
    n = x.shape[1]
    with torch.cuda.device(x.device):
        _lib.check(lib.prt_bundle_moments_async(x.device.index, n, pitch, _ptr(x), _ptr(mask), mode,
                                                _ptr(ref_dev), ref_kind, _ptr(ws.out[slot]),
                                                _ptr(ws.scratch), _stream_handle(x.device)))
    return ws.out[slot]


def spot_from_moments(m, ref):
    """(count, centroid (3,), rms spot radius about the centroid) from one-pass moments
    {n, S1, S2} taken about ``ref``: centroid = ref + S1/n, rms^2 = (sum S2 - |S1|^2/n)/(n - 1)
    -- the estimators of RayBundleAnalysis (analysis/ray_analysis.py:44-86; same 1e-17 guards)."""
    import numpy as np
    m = np.asarray(m, dtype=float)
    n = m[0]
    c = m[1:4] / (n + 1e-17)
    ss = float(np.sum(m[4:7]) - n * np.sum(c * c))
    return n, np.asarray(ref, dtype=float) + c, float(np.sqrt(max(ss, 0.0) / (n - 1 + 1e-17)))


def poynting_dir(k, e_re=None, e_im=None, default_e=False):
    """unit Poynting directions (3, N) on the device (prt_poynting_dir; ray.py:136-152)"""
    lib = _lib.load()
    (k, e_re, e_im) = [_rows_contiguous(t) for t in (k, e_re, e_im)]
    _check_rays(k, "k")
    with torch.cuda.device(k.device):
        d = torch.empty((3, k.shape[1]), dtype=torch.float64, device=k.device)
        _lib.check(lib.prt_poynting_dir(k.device.index, k.shape[1], _ptr(k), _ptr(e_re), _ptr(e_im),
                                        1 if default_e else 0, _ptr(d), _stream_handle(k.device)))
    return d


def path_sums(xs, ks=None, mode=0):
    """per-ray arc length (mode 0) / phase difference (mode 1) over a list of (3, N) device arrays"""
    lib = _lib.load()
    xs = [_rows_contiguous(t) for t in xs]
    n = xs[0].shape[1]
    dev = xs[0].device
    xt = (ctypes.c_void_p * len(xs))(*[t.data_ptr() for t in xs])
    kt = None
    if ks is not None:
        ks = [_rows_contiguous(t) for t in ks]
        kt = (ctypes.c_void_p * len(ks))(*[t.data_ptr() for t in ks])
    with torch.cuda.device(dev):
        out = torch.empty(n, dtype=torch.float64, device=dev)
        _lib.check(lib.prt_path_sums(dev.index, len(xs), n, xt, kt, mode, _ptr(out), _stream_handle(dev)))
    return out


def efield_perp(k):
    """a unit E field perpendicular to k on the device (prt_efield_perp)."""
    lib = _lib.load()
    k = _rows_contiguous(k)
    _check_rays(k, "k")
    with torch.cuda.device(k.device):
        e = torch.empty_like(k)
        _lib.check(lib.prt_efield_perp(k.device.index, k.shape[1], _ptr(k), _ptr(e),
                                       _stream_handle(k.device)))
    return e


PINNED_D2H_MIN_BYTES = 1 << 20


def stack_to_host(tensors):
    """(P, R, N) NumPy array of P equally shaped device tensors (R, N) -- the reference's array
    layout.  Large results are copied straight into ONE page-locked buffer (torch's caching host
    allocator; the returned array is a view of it), which avoids the pageable-memory bounce of
    ``.cpu()`` and the extra np.stack copy."""
    import numpy as np
    tensors = list(tensors)
    if not tensors:
        return np.zeros((0,))
    t0 = tensors[0]
    nbytes = len(tensors) * t0.numel() * t0.element_size()
    if not t0.is_cuda or nbytes < PINNED_D2H_MIN_BYTES:
        return np.stack([t.cpu().numpy() for t in tensors])
    try:
        out = torch.empty((len(tensors),) + tuple(t0.shape), dtype=t0.dtype, pin_memory=True)
    except RuntimeError:          # no page-locked memory left: pageable copies
        return np.stack([t.cpu().numpy() for t in tensors])
    for (p, t) in enumerate(tensors):
        out[p].copy_(t, non_blocking=True)
    torch.cuda.current_stream(t0.device).synchronize()
    return out.numpy()


def ray_rows(n, device):
    """an uninitialised row-pitched (3, n) array for an input bundle (rows prt_recommended_pitch(n)
    apart).  Big bundles get arena memory of a kind the path arrays do not use (``placed.InputRows``):
    input loads that share a kind of HBM with the march's write streams cost it 5 %."""
    pitch = recommended_pitch(n)
    device = torch.device(device)
    if device.type == "cuda" and 3 * pitch * 8 >= placed.PLACED_INPUT_MIN_BYTES and placed.DISABLED is None:
        try:
            with torch.cuda.device(device):
                return placed.input_rows.take(device, pitch)[:, :n]
        except _lib.PrtError as exc:
            # no whole free slabs left / no virtual-memory API: torch memory, slower placement
            if exc.code != _lib.ERR_NOMEM:
                placed.disable("prt_arena_alloc failed: %s" % exc)
    return _torch_alloc(lambda: torch.empty((3, pitch), dtype=torch.float64, device=device))[:, :n]


def to_device_rays(a, device, pitched=True):
    """numpy (3, N) real or zero-imaginary complex -> float64 device tensor.  With
    ``pitched`` the rows live in a (3, prt_recommended_pitch(N)) allocation (the returned
    tensor is its [:, :N] view), so that the fused kernel can use aligned 16-B loads for
    any N."""
    a = np.asarray(a)
    if np.iscomplexobj(a):
        a = a.real
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))
    n = t.shape[1]
    if not pitched or n == 0:
        return t.to(device)
    view = ray_rows(n, device)
    view.copy_(t)
    return view
