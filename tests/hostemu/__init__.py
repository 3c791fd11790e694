"""TEST INFRASTRUCTURE -- the sources of libprt compiled for the HOST (tests/hostemu/hip/hip_runtime.h explains what that is
and what it is not), and a NumPy driver of its C ABI.

Only ``tests/`` uses this.  Nothing under ``pyrate_amd/`` imports it, ``pyrate_amd._lib`` cannot load its library (another
path, no switch), and ``bench.py`` never sees it: the product runs on the gfx950 build of the same sources or not at all.
What the host build is for: the kernels' index arithmetic under AddressSanitizer / UBSan, and their arithmetic and
launch-site logic against the oracle and the golden vectors in a container that has no GPU.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
BUILD = os.path.join(HERE, "_build")
SOURCES = [os.path.join(HERE, "hostemu_prt.cpp"), os.path.join(HERE, "hip", "hip_runtime.h")] + \
    [os.path.join(ROOT, "pyrate_amd", "csrc", f) for f in ("prt.hip", "prt_kernels.h", "prt_device.h", "prt_aniso.h",
                                                           "prt_aniso_cplx.h", "prt_placed.h")] + \
    [os.path.join(ROOT, "include", "prt.h"), os.path.abspath(__file__)]          # (this file: the compiler flags)
CLANG_CANDIDATES = ["/opt/rocm/lib/llvm/bin/clang++", "/opt/rocm/llvm/bin/clang++", "clang++"]


def find_clang():
    import shutil
    for c in CLANG_CANDIDATES:
        p = c if os.path.isabs(c) and os.path.exists(c) else shutil.which(c)
        if p:
            return p
    return None


def _out(sanitize):
    return os.path.join(BUILD, {False: "libprt_hostemu.so", True: "libprt_hostemu_san.so",
                                "thread": "libprt_hostemu_tsan.so"}[sanitize])


def build(sanitize=False, force=False):
    """clang++ (x86-64) on tests/hostemu/hostemu_prt.cpp -> tests/hostemu/_build/libprt_hostemu[_san].so.  Same language
    level and contraction setting as the gfx950 build (pyrate_amd/build.py); -O1 (the sanitized build: + address,undefined
    with a shared runtime, see ``sanitizer_preload``)."""
    out = _out(sanitize)
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in SOURCES):
        return out
    clang = find_clang()
    if clang is None:
        raise RuntimeError("no clang++ for the host build")
    os.makedirs(BUILD, exist_ok=True)
    cmd = [clang, "-x", "c++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-I" + HERE, "-ffp-contract=on",
           "-fno-math-errno", "-Wall", "-Wno-unused-function", "-Wno-unknown-attributes"]
    if sanitize == "thread":
        cmd += ["-fsanitize=thread", "-shared-libsan", "-fno-omit-frame-pointer"]
    elif sanitize:
        cmd += ["-fsanitize=address,undefined,float-cast-overflow", "-fno-sanitize-recover=undefined,float-cast-overflow", "-shared-libsan",
                "-fno-omit-frame-pointer"]
    subprocess.run(cmd + [SOURCES[0], "-o", out + ".tmp"], check=True)
    os.replace(out + ".tmp", out)
    return out


def build_all():
    """the three builds (plain, address + undefined behaviour, thread) side by side: what a fresh checkout pays once,
    about 90 s instead of 3 minutes"""
    import concurrent.futures
    with concurrent.futures.ThreadPoolExecutor(3) as pool:
        jobs = [pool.submit(build, kind) for kind in (False, True, "thread")]
        return [j.result() for j in jobs]


def sanitizer_preload(which="asan"):
    """the shared AddressSanitizer (``which`` "tsan": ThreadSanitizer) runtime of the clang that built the sanitized
    library (LD_PRELOAD for an uninstrumented python), or None"""
    clang = find_clang()
    if clang is None:
        return None
    p = subprocess.run([clang, "-print-file-name=libclang_rt.%s-x86_64.so" % which], capture_output=True, text=True).stdout.strip()
    if which != "asan":
        return p if os.path.isabs(p) and os.path.exists(p) else None
    if not os.path.isabs(p) or not os.path.exists(p):
        p = subprocess.run([clang, "-print-file-name=libclang_rt.asan.so"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


_libs = {}


def load(path=None):
    """the host build behind the prototypes of pyrate_amd/_lib.py (the table of include/prt.h's entry points)"""
    from pyrate_amd import _lib as product          # prototypes and struct mirrors only: product.load() is never called
    path = path or os.environ.get("PRT_HOSTEMU_LIBRARY") or build()
    if path in _libs:
        return _libs[path]
    lib = ctypes.CDLL(path)
    for (name, (restype, argtypes)) in product.PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    assert lib.prt_abi_version() == 1000 + product.ABI_VERSION          # (hostemu_prt.cpp: the product refuses this build)
    assert lib.prt_sizeof_surface() == ctypes.sizeof(product.PrtSurface)
    assert lib.prt_sizeof_trace_args() == ctypes.sizeof(product.PrtTraceArgs)
    _libs[path] = lib
    return lib


class HostemuError(RuntimeError):
    def __init__(self, code, detail):
        self.code = code
        RuntimeError.__init__(self, "libprt (host build) error %d: %s" % (code, detail))


def _p(a):
    return None if a is None else a.ctypes.data


def _alloc(shape, dtype=np.float64, fill=np.nan, misalign=False):
    """a C-contiguous array of ``shape``; ``misalign``: its first element sits one element behind a 16-B boundary (doubles:
    8-B aligned only; bytes: an odd address) -- what the entry points must answer with their 8-B / 1-B access forms"""
    n = int(np.prod(shape))
    base = np.full(n + 2, fill, dtype=dtype)
    assert base.ctypes.data % 16 == 0
    view = base[1:1 + n] if misalign else base[:n]
    return view.reshape(shape)


def _rows(a, pitch, misalign=False):
    """(3, n) values -> a (3, pitch) array whose first n columns hold them (the rest NaN), C-contiguous"""
    a = np.asarray(a)
    a = np.asarray(np.real(a) if np.iscomplexobj(a) else a, dtype=np.float64)
    out = _alloc((3, pitch or a.shape[1]), misalign=misalign)
    out[:, :a.shape[1]] = a
    return out


class HostSystem(object):
    """prt_system_t of the host build; every method is one entry point of include/prt.h on NumPy arrays"""

    def __init__(self, records, lib=None):
        from pyrate_amd import _lib as product
        from pyrate_amd.surface_table import pack_table
        self.product = product
        self.lib = lib or load()
        self.records = list(records)
        self.S = len(self.records)
        self._table = pack_table(self.records)
        h = ctypes.c_void_p()
        self._check(self.lib.prt_system_create(self._table, self.S, 0, ctypes.byref(h)))
        self._h = h
        from pyrate_amd import surface_table
        self.complex_eps = surface_table.has_complex_eps(self.records)
        self.all_isotropic = all(r["material"]["type"] == "isotropic" for r in self.records) and not self.complex_eps

    def _check(self, rc):
        if rc < 0:
            raise HostemuError(rc, self.lib.prt_last_error().decode() or self.lib.prt_strerror(rc).decode())
        return rc

    def close(self):
        if self._h:
            self.lib.prt_system_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def layout(self):
        return self._check(self.lib.prt_system_layout(self._h))

    def ray_counts(self, n0):
        n_in = (ctypes.c_int64 * self.S)()
        n_out = (ctypes.c_int64 * self.S)()
        self._check(self.lib.prt_system_ray_counts(self._h, n0, n_in, n_out))
        return list(n_in), list(n_out)

    # -- whole sequence -------------------------------------------------------------------------------------------------
    def trace(self, x0, k0=None, e0_re=None, e0_im=None, mode=0, pitch=None, in_pitch=None, uniform=None,
              first_dir=None, want_nonconv=False, want_fields=False, flags=False, want_k_im=False, misalign=False):
        """prt_trace_ex -> the dense per-surface records of tests/_golden.py (x_hit, valid, k_out, valid_out [, nonconv,
        e_re]).  pitch: out_pitch (None: the library's recommendation for the layout, 0: tight); in_pitch: row pitch of
        the input arrays (None: tight); uniform = (k, e | None, kind) replaces k0 / e0 (prt.h PRT_FIRST_*_UNIFORM)."""
        P = self.product
        n0 = int(np.asarray(x0).shape[1])
        (n_in, n_out) = self.ray_counts(n0)
        surfaces = list(range(self.S))
        if mode == P.MODE_IMAGE:
            (n_in, n_out, surfaces) = (n_in[-1:], n_out[-1:], surfaces[-1:])
        if self.all_isotropic:
            if pitch is None:
                pitch = int(self.lib.prt_recommended_pitch(n0))
            Pp = pitch or n0
            rows = len(n_in)
            x_hit = _alloc((rows, 3, Pp), misalign=misalign)
            k_out = _alloc((rows, 3, Pp), misalign=misalign)
            valid = _alloc((rows, Pp), np.uint8, 7, misalign)
            valid_out = _alloc((rows, Pp), np.uint8, 7, misalign)
            nonconv = _alloc((rows, Pp), np.uint8, 0, misalign) if want_nonconv else None
        else:
            if pitch is None:
                pitch = int(self.lib.prt_crystal_pitch(n0))
            if self.layout() != P.LAYOUT_CONCATENATED_PITCHED or pitch < n0:
                pitch = 0
            if not pitch:
                in_pitch = None             # the per-surface march takes tight arrays (the engine packs them: _trace_args)
            Pp = pitch or n0
            (pin, pout) = ([c // n0 * Pp for c in n_in], [c // n0 * Pp for c in n_out])
            x_hit = np.full(3 * sum(pin), np.nan)
            k_out = np.full(3 * sum(pout), np.nan)
            valid = np.zeros(sum(pin), dtype=np.uint8)
            valid_out = np.zeros(sum(pout), dtype=np.uint8)
            nonconv = np.zeros(sum(pin), dtype=np.uint8) if want_nonconv else None
        a = P.PrtTraceArgs()
        a.struct_bytes = ctypes.sizeof(P.PrtTraceArgs)
        a.mode = mode | (P.MODE_FLAGS if flags else 0)
        a.n0 = n0
        keep = [_rows(x0, in_pitch, misalign)]
        a.x0 = _p(keep[0])
        a.in_pitch = in_pitch or 0
        if uniform is not None:
            (ku, eu, kind) = uniform
            a.k_uniform[:] = [float(v) for v in ku]
            if eu is not None:
                a.e_uniform_re[:] = [float(np.real(v)) for v in eu]
                a.e_uniform_im[:] = [float(np.imag(v)) for v in eu]
            a.first_dir = kind
        else:
            keep.append(_rows(k0, in_pitch, misalign))
            a.k0 = _p(keep[-1])
            if e0_re is not None:
                keep.append(_rows(e0_re, in_pitch, misalign))
                a.e0_re = _p(keep[-1])
            if e0_im is not None:
                keep.append(_rows(e0_im, in_pitch, misalign))
                a.e0_im = _p(keep[-1])
            a.first_dir = P.FIRST_E if first_dir is None else first_dir
        a.out_pitch = pitch
        (a.x_hit, a.k_out, a.valid) = (_p(x_hit), _p(k_out), _p(valid))
        a.valid_out = None if flags else _p(valid_out)
        a.nonconv = _p(nonconv)
        (e_re, e_im, k_im) = (None, None, None)
        if want_fields:
            e_re = np.zeros_like(k_out)
            a.e_out_re = _p(e_re)
            if self.complex_eps:
                e_im = np.zeros_like(k_out)
                a.e_out_im = _p(e_im)
        if self.complex_eps or ((want_k_im or want_fields) and not self.all_isotropic and mode == P.MODE_PATH and pitch):
            k_im = np.full_like(k_out, np.nan)
            a.k_out_im = _p(k_im)
        self._check(self.lib.prt_trace_ex(self._h, ctypes.byref(a)))
        dense = []
        if self.all_isotropic:
            for r in range(len(n_in)):
                d = dict(x_hit=x_hit[r, :, :n0].copy(), k_out=k_out[r, :, :n0].copy())
                if flags:
                    d.update(valid=(valid[r, :n0] & 1), valid_out=((valid[r, :n0] >> 1) & 1), nonconv=((valid[r, :n0] >> 2) & 1))
                else:
                    d.update(valid=valid[r, :n0].copy(), valid_out=valid_out[r, :n0].copy())
                    if want_nonconv:
                        d["nonconv"] = nonconv[r, :n0].copy()
                dense.append(d)
            # (a thread owns two adjacent rays: with an odd n0 the aligned 16-B store of the last thread covers column n0)
            n1 = n0 + (n0 % 2)
            self.padding_untouched = bool(np.all(np.isnan(x_hit[:, :, n1:])) and np.all(valid[:, n1:] == 7))
        else:
            def take(buf, counts, s_idx, comps):
                off = comps * sum(counts[:s_idx])
                B = counts[s_idx] // Pp
                blk = buf[off:off + comps * counts[s_idx]].reshape(comps, B, Pp)[:, :, :n0]
                return blk.reshape(comps, B * n0).copy()
            for r in range(len(n_in)):
                d = dict(x_hit=take(x_hit, pin, r, 3), k_out=take(k_out, pout, r, 3),
                         valid=take(valid, pin, r, 1)[0], valid_out=take(valid_out, pout, r, 1)[0])
                if want_nonconv:
                    d["nonconv"] = take(nonconv, pin, r, 1)[0]
                if want_fields:
                    d["e_re"] = take(e_re, pout, r, 3)
                    if e_im is not None:
                        d["e_im"] = take(e_im, pout, r, 3)
                if k_im is not None:
                    d["k_im"] = take(k_im, pout, r, 3)
                    d["k_out"] = d["k_out"] + 1j * d["k_im"]
                dense.append(d)
        return dense

    # -- one surface at a time (row-pitched arrays, two rays per thread) ---------------------------------------------
    def propagate_rows(self, s, x, k, direction=None, e_re=None, e_im=None, default_e=False, valid_in=None, pitch=None,
                       want_nonconv=False, misalign=False):
        n = x.shape[1]
        (xi, ki) = (_rows(x, pitch, misalign), _rows(k, pitch, misalign))
        Pp = pitch or n
        x_hit = _alloc((3, Pp), misalign=misalign)
        valid = _alloc((n,), np.uint8, 7, misalign)
        nonconv = _alloc((n,), np.uint8, 0, misalign) if want_nonconv else None
        extra = [None if t is None else _rows(t, pitch, misalign) for t in (direction, e_re, e_im)]
        vin = None if valid_in is None else np.ascontiguousarray(valid_in, dtype=np.uint8)
        self._check(self.lib.prt_propagate_rows(self._h, s, n, _p(xi), Pp, _p(ki), Pp, _p(extra[0]), _p(extra[1]),
                                                _p(extra[2]), 1 if default_e else 0, _p(vin), _p(x_hit), Pp, _p(valid),
                                                _p(nonconv), None))
        assert np.all(np.isnan(x_hit[:, n + (n % 2):]))
        return (x_hit[:, :n].copy(), valid) + ((nonconv,) if want_nonconv else ())

    def interact_rows(self, s, x_hit, k, valid_in=None, pitch=None, want_dir=False, misalign=False):
        n = x_hit.shape[1]
        (xi, ki) = (_rows(x_hit, pitch, misalign), _rows(k, pitch, misalign))
        Pp = pitch or n
        k_out = _alloc((3, Pp), misalign=misalign)
        d_out = _alloc((3, Pp), misalign=misalign) if want_dir else None
        valid_out = _alloc((n,), np.uint8, 7, misalign)
        vin = None if valid_in is None else np.ascontiguousarray(valid_in, dtype=np.uint8)
        self._check(self.lib.prt_interact_rows(self._h, s, n, _p(xi), Pp, _p(ki), Pp, _p(vin), _p(k_out), Pp, _p(d_out),
                                               _p(valid_out), None))
        assert np.all(np.isnan(k_out[:, n + (n % 2):]))
        return (k_out[:, :n].copy(), valid_out) + ((d_out[:, :n].copy(),) if want_dir else ())

    def surface_step_rows(self, s, x, k, direction=None, e_re=None, e_im=None, default_e=False, valid_in=None,
                          pitch=None, want_nonconv=False, misalign=False):
        n = x.shape[1]
        (xi, ki) = (_rows(x, pitch, misalign), _rows(k, pitch, misalign))
        Pp = pitch or n
        x_hit = _alloc((3, Pp), misalign=misalign)
        k_out = _alloc((3, Pp), misalign=misalign)
        valid = _alloc((n,), np.uint8, 7, misalign)
        valid_out = _alloc((n,), np.uint8, 7, misalign)
        nonconv = _alloc((n,), np.uint8, 0, misalign) if want_nonconv else None
        extra = [None if t is None else _rows(t, pitch, misalign) for t in (direction, e_re, e_im)]
        vin = None if valid_in is None else np.ascontiguousarray(valid_in, dtype=np.uint8)
        self._check(self.lib.prt_surface_step_rows(self._h, s, n, _p(xi), Pp, _p(ki), Pp, _p(extra[0]), _p(extra[1]),
                                                   _p(extra[2]), 1 if default_e else 0, _p(vin), _p(x_hit), _p(k_out), Pp,
                                                   _p(valid), _p(valid_out), _p(nonconv), None))
        assert np.all(np.isnan(x_hit[:, n + (n % 2):])) and np.all(np.isnan(k_out[:, n + (n % 2):]))
        return (x_hit[:, :n].copy(), k_out[:, :n].copy(), valid, valid_out) + ((nonconv,) if want_nonconv else ())

    # -- tight arrays, one ray per thread ---------------------------------------------------------------------------------
    def propagate(self, s, x, k, direction=None, e_re=None, e_im=None, default_e=False, valid_in=None):
        n = x.shape[1]
        arrs = [None if t is None else np.ascontiguousarray(t, dtype=np.float64) for t in (x, k, direction, e_re, e_im)]
        x_hit = np.full((3, n), np.nan)
        valid = np.full(n, 7, dtype=np.uint8)
        vin = None if valid_in is None else np.ascontiguousarray(valid_in, dtype=np.uint8)
        self._check(self.lib.prt_propagate(self._h, s, n, _p(arrs[0]), _p(arrs[1]), _p(arrs[2]), _p(arrs[3]), _p(arrs[4]),
                                           1 if default_e else 0, _p(vin), _p(x_hit), _p(valid), None, None))
        return (x_hit, valid)

    def shape_eval(self, s, x, y):
        n = len(x)
        (xa, ya) = (np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(y, dtype=np.float64))
        sag = np.full(n, np.nan)
        grad = np.full((3, n), np.nan)
        self._check(self.lib.prt_shape_eval(self._h, s, n, _p(xa), _p(ya), _p(sag), _p(grad), None))
        return (sag, grad)
