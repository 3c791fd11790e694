"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes front end of oracle/seqtrace_c.c (C / OpenMP
restatement of the Conic + isotropic trace): build helper, table flattening, ``trace``
with the same return structure as ``oracle.seqtrace_np.trace``.
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
REC = 40
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


def load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(OUT)
        lib.seqtrace_c.restype = ctypes.c_int
        lib.seqtrace_c.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64] + [ctypes.c_void_p] * 7 + \
            [ctypes.c_int]
        lib.seqtrace_c_threads.restype = ctypes.c_int
        _lib = lib
    return _lib


def flat_table(records):
    tab = np.zeros((len(records), REC))
    for (s, rec) in enumerate(records):
        if rec["shape"]["type"] != "conic" or rec["material"]["type"] != "isotropic":
            raise ValueError("the C oracle covers Conic shapes and isotropic media only")
        tab[s, 0] = rec["shape"]["curv"]
        tab[s, 1] = rec["shape"]["cc"]
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
        tab[s, 30] = rec["material"]["n"]
        tab[s, 31:40] = np.asarray(rec["B_mat"]).reshape(9)
    return tab


class Workspace(object):
    """pre-allocated (and pre-touched) output arrays, so that repeated timed calls do not pay
    first-touch page faults inside the parallel region"""

    def __init__(self, n_surfaces, n):
        self.x_hit = np.zeros((n_surfaces, 3, n))
        self.k_out = np.zeros((n_surfaces, 3, n))
        self.valid = np.zeros((n_surfaces, n), dtype=np.uint8)
        self.valid_out = np.zeros((n_surfaces, n), dtype=np.uint8)


def trace_arrays(records, x0, k0, E0=None, nthreads=0, workspace=None):
    """dense arrays: x_hit (S,3,N), k_out (S,3,N), valid (S,N), valid_out (S,N), threads used"""
    lib = load()
    tab = np.ascontiguousarray(flat_table(records))
    x0 = np.ascontiguousarray(np.real(x0), dtype=np.float64)
    k0 = np.ascontiguousarray(np.real(k0), dtype=np.float64)
    n = x0.shape[1]
    if E0 is None:
        E0 = np.zeros((3, n))
        E0[1, :] = 1.
    with np.errstate(invalid="ignore", divide="ignore"):
        d0 = np.ascontiguousarray(seqtrace_np.poynting_direction(k0, np.asarray(E0)))   # ray.py:136-152
    S = len(records)
    ws = workspace if workspace is not None else Workspace(S, n)
    (x_hit, k_out, valid, valid_out) = (ws.x_hit, ws.k_out, ws.valid, ws.valid_out)
    assert x_hit.shape == (S, 3, n)
    used = lib.seqtrace_c(tab.ctypes.data, S, n, x0.ctypes.data, k0.ctypes.data, d0.ctypes.data,
                          x_hit.ctypes.data, k_out.ctypes.data, valid.ctypes.data, valid_out.ctypes.data,
                          int(nthreads))
    return x_hit, k_out, valid, valid_out, used


def trace(records, x0, k0, E0=None, nthreads=0):
    """same structure as seqtrace_np.trace (list of per-surface dicts)"""
    (x_hit, k_out, valid, valid_out, _) = trace_arrays(records, x0, k0, E0, nthreads)
    n = x_hit.shape[2]
    return [dict(x_hit=x_hit[s], valid=valid[s].astype(bool), k_out=k_out[s],
                 valid_out=valid_out[s].astype(bool), ray_id=np.arange(n), E_out=None)
            for s in range(len(records))]
