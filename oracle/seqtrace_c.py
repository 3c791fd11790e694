"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes front end of oracle/seqtrace_c.c (C / OpenMP
restatement of the reference's trace: Conic / Asphere / XYPolynomials / Biconic shapes, isotropic and
anisotropic media): build helper, table flattening, ``trace`` with the same return structure as
``oracle.seqtrace_np.trace``.

The anisotropic interface solves the reference's 6x6 pencil with LAPACK's zggev -- the routine behind the
reference's ``scipy.linalg.eig`` call (raytracer/material/material.py:435).  The C code does not link a LAPACK:
the entry point is taken from SciPy's own (``scipy.linalg.cython_lapack.__pyx_capi__["zggev"]``) and handed to
the library, so the restatement runs the very routine the reference runs.
"""
import ctypes
import os
import shutil
import subprocess

import numpy as np

from . import seqtrace_np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "seqtrace_c.c")
OUT = os.path.join(HERE, "libseqtrace_c.so")
OUT_SANITIZED = os.path.join(HERE, "libseqtrace_c_san.so")
# PRT_ORACLE_C_LIBRARY: load this build of the library instead (tests/test_oracle_c.py: the sanitizer build)
REC = 72
SHAPES = {"conic": 0, "asphere": 1, "xypoly": 2, "biconic": 3}
_lib = None


def build(force=False):
    """gcc -O2 -fno-math-errno -fopenmp -shared -fPIC seqtrace_c.c -> oracle/libseqtrace_c.so
    (baseline x86-64 code: the prebuilt .so travels to the GPU box, whose CPU differs)"""
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    gcc = shutil.which("gcc")
    if gcc is None:
        if os.path.exists(OUT):
            return OUT
        raise RuntimeError("gcc not found and %s is not built" % OUT)
    subprocess.run([gcc, "-O2", "-fno-math-errno", "-fopenmp", "-shared", "-fPIC", "-o", OUT, SRC, "-lm"],
                   check=True)
    return OUT


def build_sanitized(force=False):
    """the same source with AddressSanitizer + UndefinedBehaviorSanitizer: gcc -O1 -g
    -fsanitize=address,undefined -> oracle/libseqtrace_c_san.so.  Loaded into an uninstrumented python only with
    libasan preloaded (``sanitizer_preload``); tests/test_oracle_c.py runs golden cases under it in a subprocess."""
    if not force and os.path.exists(OUT_SANITIZED) and os.path.getmtime(OUT_SANITIZED) >= os.path.getmtime(SRC):
        return OUT_SANITIZED
    gcc = shutil.which("gcc")
    if gcc is None:
        raise RuntimeError("gcc not found")
    subprocess.run([gcc, "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
                    "-fno-sanitize-recover=undefined", "-fno-math-errno", "-fopenmp", "-shared", "-fPIC",
                    "-o", OUT_SANITIZED, SRC, "-lm"], check=True)
    return OUT_SANITIZED


def sanitizer_preload():
    """path of gcc's libasan.so (what LD_PRELOAD needs for the sanitized library), or None"""
    gcc = shutil.which("gcc")
    if gcc is None:
        return None
    path = subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return os.path.realpath(path) if os.path.isabs(path) and os.path.exists(path) else None


def _zggev_pointer():
    """address of LAPACK's zggev inside SciPy (None if this SciPy does not export it)"""
    try:
        from scipy.linalg import cython_lapack
        cap = cython_lapack.__pyx_capi__["zggev"]
        get = ctypes.pythonapi.PyCapsule_GetPointer
        get.restype = ctypes.c_void_p
        get.argtypes = [ctypes.py_object, ctypes.c_char_p]
        name = ctypes.pythonapi.PyCapsule_GetName
        name.restype = ctypes.c_char_p
        name.argtypes = [ctypes.py_object]
        return get(cap, name(cap))
    except Exception:
        return None


def load():
    global _lib
    if _lib is None:
        path = os.environ.get("PRT_ORACLE_C_LIBRARY")
        if not path:
            path = build()
        lib = ctypes.CDLL(path)
        lib.seqtrace_c.restype = ctypes.c_int
        lib.seqtrace_c.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64] + \
            [ctypes.c_void_p] * 7 + [ctypes.c_int]
        lib.seqtrace_c_general.restype = ctypes.c_int
        lib.seqtrace_c_general.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64] + \
            [ctypes.c_void_p] * 10 + [ctypes.c_int]
        lib.seqtrace_c_threads.restype = ctypes.c_int
        lib.seqtrace_c_set_zggev.argtypes = [ctypes.c_void_p]
        lib.seqtrace_c_has_zggev.restype = ctypes.c_int
        ptr = _zggev_pointer()
        if ptr:
            lib.seqtrace_c_set_zggev(ptr)
        _lib = lib
    return _lib


def supports(records):
    """can the C oracle trace this table?  (shapes: conic, asphere, xypoly, biconic; crystals need SciPy's zggev)"""
    crystals = False
    for rec in records:
        if rec["shape"]["type"] not in SHAPES:
            return False
        if rec["material"]["type"] == "anisotropic":
            crystals = True
        elif rec["material"]["type"] != "isotropic":
            return False
    return (not crystals) or bool(load().seqtrace_c_has_zggev())


def flat_table(records):
    """(table (S, REC), coefficient array)"""
    tab = np.zeros((len(records), REC))
    coeffs = []
    for (s, rec) in enumerate(records):
        sh = rec["shape"]
        if sh["type"] not in SHAPES:
            raise ValueError("the C oracle covers Conic, Asphere, XYPolynomials and Biconic shapes only")
        tab[s, 40] = SHAPES[sh["type"]]
        tab[s, 42] = len(coeffs)
        if sh["type"] in ("conic", "asphere"):
            tab[s, 0] = sh["curv"]
            tab[s, 1] = sh["cc"]
            if sh["type"] == "asphere":
                tab[s, 41] = len(sh["coeffs"])
                coeffs += [float(a) for a in sh["coeffs"]]
        elif sh["type"] == "xypoly":
            tab[s, 41] = len(sh["terms"])
            tab[s, 64] = sh["normradius"]
            for (i, j, c) in sh["terms"]:
                coeffs += [float(i), float(j), float(c)]
        else:
            (tab[s, 0], tab[s, 1], tab[s, 62], tab[s, 63]) = (sh["curvx"], sh["ccx"], sh["curvy"], sh["ccy"])
            tab[s, 41] = len(sh["coeffs"])
            for (a, b) in sh["coeffs"]:
                coeffs += [float(a), float(b)]
        tab[s, 2:11] = np.asarray(rec["B_shape"]).reshape(9)
        tab[s, 11:14] = rec["g_shape"]
        ap = rec["aperture"]
        if ap["type"] == "circular":
            tab[s, 14:17] = (1, ap["minradius"], ap["maxradius"])
        elif ap["type"] == "rectangular":
            tab[s, 14:17] = (2, ap["width"], ap["height"])
        tab[s, 17:26] = np.asarray(rec["B_ap"]).reshape(9)
        tab[s, 26:29] = rec["g_ap"]
        tab[s, 29] = 1.0 if rec["interaction"] == "mirror" else 0.0
        mat = rec["material"]
        if mat["type"] == "isotropic":
            tab[s, 30] = mat["n"]
            tab[s, 53] = mat.get("n_im", 0.0)          # absorbing isotropic medium (behind the last surface)
        elif mat["type"] == "anisotropic":
            tab[s, 43] = 1
            tab[s, 44:53] = np.asarray(mat["eps_re"], dtype=float).reshape(9)
            tab[s, 53:62] = np.asarray(mat["eps_im"], dtype=float).reshape(9)
        else:
            raise ValueError("the C oracle covers isotropic and anisotropic media only")
        tab[s, 31:40] = np.asarray(rec["B_mat"]).reshape(9)
    return tab, np.asarray(coeffs + [0.0], dtype=np.float64)


def ray_counts(records, n):
    """rays entering / leaving every surface (anisotropic interfaces double the count)"""
    (n_in, n_out) = ([], [])
    for rec in records:
        n_in.append(n)
        if rec["material"]["type"] == "anisotropic":
            n *= 2
        n_out.append(n)
    return n_in, n_out


class Workspace(object):
    """pre-allocated (and pre-touched) output arrays, so that repeated timed calls do not pay
    first-touch page faults inside the parallel region.  All-isotropic tables: (S,3,n) / (S,n) arrays;
    tables with crystals: the concatenated layout (plus imaginary parts of k and the E fields)."""

    def __init__(self, records, n):
        S = len(records)
        # (general path: crystals, or any absorbing medium -- complex wave vectors)
        self.general = any(r["material"]["type"] == "anisotropic" or r["material"].get("n_im", 0.0) != 0.0
                           for r in records)
        if not self.general:
            self.x_hit = np.zeros((S, 3, n))
            self.k_out = np.zeros((S, 3, n))
            self.valid = np.zeros((S, n), dtype=np.uint8)
            self.valid_out = np.zeros((S, n), dtype=np.uint8)
        else:
            (self.n_in, self.n_out) = ray_counts(records, n)
            self.x_hit = np.zeros(3 * sum(self.n_in))
            self.k_re = np.zeros(3 * sum(self.n_out))
            self.k_im = np.zeros(3 * sum(self.n_out))
            self.e_re = np.zeros(3 * sum(self.n_out))
            self.e_im = np.zeros(3 * sum(self.n_out))
            self.valid = np.zeros(sum(self.n_in), dtype=np.uint8)
            self.valid_out = np.zeros(sum(self.n_out), dtype=np.uint8)


def trace_arrays(records, x0, k0, E0=None, nthreads=0, workspace=None):
    """All-isotropic tables: dense arrays x_hit (S,3,N), k_out (S,3,N), valid (S,N), valid_out (S,N), threads used.
    Tables with crystals: (workspace, threads used) -- the concatenated arrays live in the Workspace."""
    lib = load()
    (tab, cf) = flat_table(records)
    tab = np.ascontiguousarray(tab)
    x0 = np.ascontiguousarray(np.real(x0), dtype=np.float64)
    k0 = np.ascontiguousarray(np.real(k0), dtype=np.float64)
    n = x0.shape[1]
    if E0 is None:
        E0 = np.zeros((3, n))
        E0[1, :] = 1.
    with np.errstate(invalid="ignore", divide="ignore"):
        d0 = np.ascontiguousarray(seqtrace_np.poynting_direction(k0, np.asarray(E0)))   # ray.py:136-152
    S = len(records)
    ws = workspace if workspace is not None else Workspace(records, n)
    if ws.general:
        if not lib.seqtrace_c_has_zggev():
            raise RuntimeError("the C oracle needs SciPy's zggev for anisotropic media")
        used = lib.seqtrace_c_general(tab.ctypes.data, cf.ctypes.data, S, n, x0.ctypes.data, k0.ctypes.data,
                                      d0.ctypes.data, ws.x_hit.ctypes.data, ws.k_re.ctypes.data, ws.k_im.ctypes.data,
                                      ws.e_re.ctypes.data, ws.e_im.ctypes.data, ws.valid.ctypes.data,
                                      ws.valid_out.ctypes.data, int(nthreads))
        if used < 0:
            raise MemoryError("seqtrace_c_general")
        return ws, used
    (x_hit, k_out, valid, valid_out) = (ws.x_hit, ws.k_out, ws.valid, ws.valid_out)
    assert x_hit.shape == (S, 3, n)
    used = lib.seqtrace_c(tab.ctypes.data, cf.ctypes.data, S, n, x0.ctypes.data, k0.ctypes.data, d0.ctypes.data,
                          x_hit.ctypes.data, k_out.ctypes.data, valid.ctypes.data, valid_out.ctypes.data,
                          int(nthreads))
    return x_hit, k_out, valid, valid_out, used


def trace(records, x0, k0, E0=None, nthreads=0):
    """same structure as seqtrace_np.trace (list of per-surface dicts)"""
    res = trace_arrays(records, x0, k0, E0, nthreads)
    n = np.asarray(x0).shape[1]
    if len(res) == 2:
        ws = res[0]
        out = []
        (oi, oo) = (0, 0)
        ids = np.arange(n)
        for (s, rec) in enumerate(records):
            (ni, no) = (ws.n_in[s], ws.n_out[s])
            aniso = rec["material"]["type"] == "anisotropic"
            k = ws.k_re[3 * oo:3 * (oo + no)].reshape(3, no) + 1j * ws.k_im[3 * oo:3 * (oo + no)].reshape(3, no)
            e = None
            if aniso:
                ids = np.hstack((ids, ids))
                e = ws.e_re[3 * oo:3 * (oo + no)].reshape(3, no) + 1j * ws.e_im[3 * oo:3 * (oo + no)].reshape(3, no)
            out.append(dict(x_hit=ws.x_hit[3 * oi:3 * (oi + ni)].reshape(3, ni).copy(),
                            valid=ws.valid[oi:oi + ni].astype(bool),
                            k_out=k if (aniso or np.any(np.imag(k) != 0.0)) else np.real(k),
                            valid_out=ws.valid_out[oo:oo + no].astype(bool), ray_id=ids, E_out=e))
            oi += ni
            oo += no
        return out
    (x_hit, k_out, valid, valid_out, _) = res
    return [dict(x_hit=x_hit[s], valid=valid[s].astype(bool), k_out=k_out[s],
                 valid_out=valid_out[s].astype(bool), ray_id=np.arange(n), E_out=None)
            for s in range(len(records))]
