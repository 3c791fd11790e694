"""
ctypes binding of ``libprt.so`` (C ABI: ``include/prt.h``).

The library is built in-tree (``pyrate_amd/csrc/libprt.so``) by
``pyrate_amd.build.build_all()`` / ``__graft_entry__.build()``.  There is no
CPU fallback: if the shared object is missing, stale or cannot be loaded,
``load()`` raises ``ImportError`` and every entry point of the product fails
loudly.
"""
import ctypes
import os

from .surface_table import PrtSurface

_HERE = os.path.dirname(os.path.abspath(__file__))
# PRT_LIBRARY: an alternative build of the same ABI (A/B experiments); default: the in-tree library
LIB_PATH = os.environ.get("PRT_LIBRARY") or os.path.join(_HERE, "csrc", "libprt.so")

PRT_OK = 0
ERR_INVALID_ARG = -1   # PRT_ERR_INVALID_ARG
ERR_UNSUPPORTED = -2   # PRT_ERR_UNSUPPORTED
ERR_DEVICE = -3        # PRT_ERR_DEVICE
ERR_NOMEM = -5         # PRT_ERR_NOMEM
MODE_PATH = 0
MODE_IMAGE = 1
MODE_FLAGS = 2          # OR-ed in: both masks of a record in one byte (include/prt.h)
(LAYOUT_ROW_PITCHED, LAYOUT_CONCATENATED_PITCHED, LAYOUT_CONCATENATED_TIGHT) = (0, 1, 2)      # prt_system_layout

c_double_p = ctypes.c_void_p      # device pointers travel as raw addresses
c_u8_p = ctypes.c_void_p
c_stream = ctypes.c_void_p

ABI_VERSION = 7          # PRT_ABI_VERSION of include/prt.h

# name -> (restype, argtypes); must list every symbol declared in include/prt.h
PROTOTYPES = {
    "prt_abi_version": (ctypes.c_int32, []),
    "prt_device_count": (ctypes.c_int32, []),
    "prt_strerror": (ctypes.c_char_p, [ctypes.c_int32]),
    "prt_last_error": (ctypes.c_char_p, []),
    "prt_sizeof_surface": (ctypes.c_int32, []),
    "prt_system_create": (ctypes.c_int32, [ctypes.POINTER(PrtSurface), ctypes.c_int32,
                                           ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p)]),
    "prt_system_destroy": (ctypes.c_int32, [ctypes.c_void_p]),
    "prt_system_update": (ctypes.c_int32, [ctypes.c_void_p, ctypes.POINTER(PrtSurface), ctypes.c_int32, ctypes.c_void_p]),
    "prt_system_num_surfaces": (ctypes.c_int32, [ctypes.c_void_p]),
    "prt_system_layout": (ctypes.c_int32, [ctypes.c_void_p]),
    "prt_system_ray_counts": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int64,
                                               ctypes.POINTER(ctypes.c_int64),
                                               ctypes.POINTER(ctypes.c_int64)]),
    "prt_recommended_pitch": (ctypes.c_int64, [ctypes.c_int64]),
    "prt_crystal_pitch": (ctypes.c_int64, [ctypes.c_int64]),
    "prt_trace": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, c_double_p,
                                   c_double_p, c_double_p, c_double_p, ctypes.c_int32,
                                   ctypes.c_int64, c_double_p, c_double_p, c_u8_p, c_u8_p, c_u8_p,
                                   c_stream]),
    "prt_trace_timed": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                         c_double_p, c_double_p, c_double_p, c_double_p,
                                         ctypes.c_int32, ctypes.c_int64, c_double_p, c_double_p,
                                         c_u8_p, c_u8_p, c_stream, ctypes.c_int32,
                                         ctypes.POINTER(ctypes.c_double)]),
    "prt_propagate": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64,
                                       c_double_p, c_double_p, c_double_p, c_double_p,
                                       c_double_p, ctypes.c_int32, c_u8_p, c_double_p, c_u8_p, c_u8_p,
                                       c_stream]),
    "prt_interact": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64,
                                      c_double_p, c_double_p, c_u8_p, c_double_p, c_double_p,
                                      c_double_p, c_double_p, c_u8_p, c_stream]),
    "prt_propagate_rows": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, c_double_p, ctypes.c_int64,
                                            c_double_p, ctypes.c_int64, c_double_p, c_double_p, c_double_p, ctypes.c_int32,
                                            c_u8_p, c_double_p, ctypes.c_int64, c_u8_p, c_u8_p, c_stream]),
    "prt_surface_step_rows": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, c_double_p, ctypes.c_int64,
                                               c_double_p, ctypes.c_int64, c_double_p, c_double_p, c_double_p, ctypes.c_int32,
                                               c_u8_p, c_double_p, c_double_p, ctypes.c_int64, c_u8_p, c_u8_p, c_u8_p,
                                               c_stream]),
    "prt_interact_rows": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, c_double_p, ctypes.c_int64,
                                           c_double_p, ctypes.c_int64, c_u8_p, c_double_p, ctypes.c_int64, c_double_p,
                                           c_u8_p, c_stream]),
    "prt_interact_cplx": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, c_double_p, c_double_p,
                                           c_double_p, c_u8_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                           c_double_p, c_u8_p, c_stream]),
    "prt_shape_eval": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64,
                                        c_double_p, c_double_p, c_double_p, c_double_p,
                                        c_stream]),
    "prt_efield_perp": (ctypes.c_int32, [ctypes.c_int32, ctypes.c_int64, c_double_p, c_double_p,
                                         c_stream]),
    "prt_bundle_moments": (ctypes.c_int32, [ctypes.c_int32, ctypes.c_int64, ctypes.c_int64,
                                            c_double_p, c_u8_p, ctypes.c_int32,
                                            ctypes.POINTER(ctypes.c_double),
                                            ctypes.POINTER(ctypes.c_double), c_stream]),
    "prt_compact_scratch_bytes": (ctypes.c_int64, [ctypes.c_int64]),
    "prt_compact": (ctypes.c_int32, [ctypes.c_int64, c_u8_p, ctypes.c_int32,
                                     ctypes.POINTER(ctypes.c_void_p),
                                     ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p,
                                     ctypes.c_void_p, c_u8_p, c_u8_p, ctypes.c_void_p,
                                     ctypes.POINTER(ctypes.c_int64), c_stream]),
}

class PrtCollimated(ctypes.Structure):
    """ctypes mirror of prt_collimated_t"""
    _fields_ = [("radius", ctypes.c_double), ("startx", ctypes.c_double), ("starty", ctypes.c_double),
                ("startz", ctypes.c_double), ("k", ctypes.c_double * 3), ("e", ctypes.c_double * 3)]


PROTOTYPES["prt_rect_grid_count"] = (ctypes.c_int32, [ctypes.c_int32, ctypes.c_int64,
                                                     ctypes.POINTER(ctypes.c_int64),
                                                     ctypes.POINTER(ctypes.c_int64), c_stream])
PROTOTYPES["prt_collimated_bundle"] = (ctypes.c_int32, [ctypes.c_int32, ctypes.c_int64, ctypes.c_int64,
                                                       ctypes.c_int64, ctypes.POINTER(PrtCollimated),
                                                       ctypes.c_int64, c_double_p, c_double_p,
                                                       c_double_p, c_stream])

class PrtRaster(ctypes.Structure):
    """ctypes mirror of prt_raster_t"""
    _fields_ = [("ni", ctypes.c_int64), ("nj", ctypes.c_int64), ("xa", ctypes.c_void_p), ("xb", ctypes.c_void_p),
                ("ya", ctypes.c_void_p), ("yb", ctypes.c_void_p), ("clip", ctypes.c_int32), ("pad_", ctypes.c_int32)]


class PrtBundle(ctypes.Structure):
    """ctypes mirror of prt_bundle_t"""
    _fields_ = [("kind", ctypes.c_int32), ("pad_", ctypes.c_int32), ("radius", ctypes.c_double),
                ("start", ctypes.c_double * 3), ("anglex", ctypes.c_double), ("angley", ctypes.c_double),
                ("index", ctypes.c_double), ("k", ctypes.c_double * 3), ("e", ctypes.c_double * 3)]


PROTOTYPES["prt_raster_count"] = (ctypes.c_int32, [ctypes.c_int32, ctypes.POINTER(PrtRaster),
                                                  ctypes.POINTER(ctypes.c_int64), c_stream])
PROTOTYPES["prt_raster_bundle"] = (ctypes.c_int32, [ctypes.c_int32, ctypes.POINTER(PrtRaster), ctypes.c_int64,
                                                   ctypes.c_int64, ctypes.POINTER(PrtBundle), ctypes.c_int64,
                                                   c_double_p, c_double_p, c_double_p, c_double_p, c_stream])

PROTOTYPES["prt_poynting_dir"] = (ctypes.c_int32, [ctypes.c_int32, ctypes.c_int64, c_double_p, c_double_p,
                                                  c_double_p, ctypes.c_int32, c_double_p, c_stream])
PROTOTYPES["prt_path_sums"] = (ctypes.c_int32, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int64,
                                               ctypes.POINTER(ctypes.c_void_p),
                                               ctypes.POINTER(ctypes.c_void_p), ctypes.c_int32,
                                               c_double_p, c_stream])

PROTOTYPES["prt_moments_scratch_doubles"] = (ctypes.c_int64, [ctypes.c_int64])
PROTOTYPES["prt_bundle_moments_async"] = (ctypes.c_int32, [ctypes.c_int32, ctypes.c_int64, ctypes.c_int64,
                                                          c_double_p, c_u8_p, ctypes.c_int32, c_double_p,
                                                          ctypes.c_int32, c_double_p, c_double_p, c_stream])

PROTOTYPES["prt_trace_seq"] = (ctypes.c_int32, [ctypes.POINTER(PrtSurface), ctypes.c_int32, ctypes.c_int64, c_double_p,
                                               c_double_p, c_double_p, ctypes.c_void_p, ctypes.c_int32, c_double_p,
                                               c_double_p, c_u8_p, c_u8_p, ctypes.c_int32, c_stream])
PROTOTYPES["prt_trace_fields"] = (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int64, c_double_p, c_double_p,
                                                  c_double_p, c_double_p, ctypes.c_int32, c_double_p,
                                                  c_double_p, c_double_p, c_double_p, c_u8_p, c_u8_p,
                                                  c_stream])

PROTOTYPES["prt_trace_moments_scratch_doubles"] = (ctypes.c_int64, [ctypes.c_int64])
PROTOTYPES["prt_trace_moments"] = (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, c_double_p,
                                                   c_double_p, c_double_p, c_double_p, ctypes.c_int32,
                                                   ctypes.c_int64, c_double_p, c_double_p, c_u8_p, c_u8_p,
                                                   c_double_p, c_double_p, c_double_p, c_stream])

# first_dir of prt_trace_args_t (include/prt.h PRT_FIRST_*)
FIRST_E = 0
FIRST_K = 1
FIRST_DIR = 2
FIRST_E_UNIFORM = 3
FIRST_DIR_UNIFORM = 4


class PrtTraceArgs(ctypes.Structure):
    """ctypes mirror of prt_trace_args_t (include/prt.h)"""
    _fields_ = [("struct_bytes", ctypes.c_int32), ("mode", ctypes.c_int32), ("n0", ctypes.c_int64),
                ("in_pitch", ctypes.c_int64), ("x0", ctypes.c_void_p), ("k0", ctypes.c_void_p),
                ("e0_re", ctypes.c_void_p), ("e0_im", ctypes.c_void_p),
                ("first_dir", ctypes.c_int32), ("pad0_", ctypes.c_int32),
                ("k_uniform", ctypes.c_double * 3), ("e_uniform_re", ctypes.c_double * 3),
                ("e_uniform_im", ctypes.c_double * 3),
                ("out_pitch", ctypes.c_int64), ("x_hit", ctypes.c_void_p), ("k_out", ctypes.c_void_p),
                ("valid", ctypes.c_void_p), ("valid_out", ctypes.c_void_p), ("nonconv", ctypes.c_void_p),
                ("e_out_re", ctypes.c_void_p), ("e_out_im", ctypes.c_void_p), ("k_out_im", ctypes.c_void_p),
                ("x_img", ctypes.c_void_p), ("k_img", ctypes.c_void_p), ("valid_img", ctypes.c_void_p),
                ("valid_out_img", ctypes.c_void_p), ("img_pitch", ctypes.c_int64),
                ("moments_ref3", ctypes.POINTER(ctypes.c_double)), ("moments_out7_dev", ctypes.c_void_p),
                ("moments_scratch_dev", ctypes.c_void_p),
                ("timed_iters", ctypes.c_int32), ("pad1_", ctypes.c_int32),
                ("ms_avg", ctypes.POINTER(ctypes.c_double)), ("stream", ctypes.c_void_p)]


PROTOTYPES["prt_sizeof_trace_args"] = (ctypes.c_int32, [])
PROTOTYPES["prt_trace_ex"] = (ctypes.c_int32, [ctypes.c_void_p, ctypes.POINTER(PrtTraceArgs)])

PROTOTYPES["prt_arena_create"] = (ctypes.c_int32, [ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p)])
PROTOTYPES["prt_arena_destroy"] = (ctypes.c_int32, [ctypes.c_void_p])
PROTOTYPES["prt_arena_alloc"] = (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int64),
                                                 ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int32),
                                                 ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_stream])
PROTOTYPES["prt_arena_free"] = (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, c_stream])
PROTOTYPES["prt_arena_trim"] = (ctypes.c_int32, [ctypes.c_void_p])
PROTOTYPES["prt_arena_set_budget"] = (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int64])
PROTOTYPES["prt_arena_kind_of"] = (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32)])
PROTOTYPES["prt_arena_stats"] = (ctypes.c_int32, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_int32,
                                                 ctypes.POINTER(ctypes.c_double), ctypes.c_int32])
PROTOTYPES["prt_arena_note"] = (ctypes.c_char_p, [ctypes.c_void_p])

_lib = None


class PrtError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        RuntimeError.__init__(self, "libprt error %d: %s" % (code, detail))


def load():
    """Load libprt.so (once).  Raises ImportError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own HIP runtime (torch/lib/libamdhip64.so).  It must be in the process
    # BEFORE libprt.so is loaded, so that libprt's NEEDED libamdhip64.so.7 resolves to that
    # same runtime: device pointers, streams and events are then shared between the torch
    # allocator and the kernels.  Loading libprt first pulls in /opt/rocm's runtime, which
    # then coexists with torch's and reports "no ROCm-capable device".
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "pyrate_amd: %s is missing -- the HIP engine is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "There is no CPU fallback." % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as exc:
        raise ImportError("pyrate_amd: cannot load %s: %s" % (LIB_PATH, exc))
    for (name, (restype, argtypes)) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise ImportError("pyrate_amd: %s does not export %s (stale build?)"
                              % (LIB_PATH, name))
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.prt_abi_version() != ABI_VERSION:
        raise ImportError("pyrate_amd: %s has ABI version %d, this package needs %d (stale build?)"
                          % (LIB_PATH, lib.prt_abi_version(), ABI_VERSION))
    if lib.prt_sizeof_surface() != ctypes.sizeof(PrtSurface):
        raise ImportError("pyrate_amd: prt_surface_t layout mismatch: C %d bytes, ctypes %d"
                          % (lib.prt_sizeof_surface(), ctypes.sizeof(PrtSurface)))
    if lib.prt_sizeof_trace_args() != ctypes.sizeof(PrtTraceArgs):
        raise ImportError("pyrate_amd: prt_trace_args_t layout mismatch: C %d bytes, ctypes %d"
                          % (lib.prt_sizeof_trace_args(), ctypes.sizeof(PrtTraceArgs)))
    _lib = lib
    return lib


def check(code):
    """Turn a negative return code into PrtError (structural misuse raises,
    like the reference's bare Exceptions; per-ray failures never do)."""
    if code < 0:
        lib = load()
        detail = lib.prt_last_error().decode() or lib.prt_strerror(code).decode()
        raise PrtError(code, detail)
    return code
