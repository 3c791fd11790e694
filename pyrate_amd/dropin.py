"""
Entry points for using the engine from UNMODIFIED reference objects:
the reference's ``OpticalSystem`` / ``OpticalElement`` / ``Surface`` / ``Shape`` / ``Material`` /
``LocalCoordinates`` instances are only *read* (``surface_table.flatten_sequence`` is duck-typed on
the attributes the reference's own trace loop uses); rays and results live in this package's
device-resident ``RayBundle`` / ``RayPath``.

    from pyrate_amd import dropin
    rpaths = dropin.seqtrace(s, initialbundle, seq)        # s: pyrateoptics OpticalSystem
    x_img = rpaths[0].raybundles[-1].x[-1]                 # reference shapes, lazily copied

Sequences through ``AnisotropicMaterial`` are traced the same way (ray doubling in the dense path
arrays); with ``splitup=True`` the forked paths (one RayPath per branch) are carved out of that one dense
trace -- path p follows solution bit j of p at the j-th crystal interface, the order
``OpticalElement.seqtrace`` builds (optical_element.py:360-375).
"""
from .raytracer.optical_system import MAX_FUSED_CRYSTALS, seqtrace_fused, _seqtrace_fused_crystal
from .raytracer.ray import RayBundle
from .surface_table import UnsupportedError, flatten_sequence, has_complex_eps


def as_device_bundle(bundle, device=None):
    """reference RayBundle (NumPy, (P,3,N)) -> device RayBundle holding its last point"""
    if isinstance(bundle, RayBundle):
        return bundle
    return RayBundle(x0=bundle.x[-1], k0=bundle.k[-1], Efield0=bundle.Efield[-1],
                     rayID=bundle.rayID, wave=bundle.wave, device=device)


def seqtrace(system, initialbundle, elementsequence, splitup=False, device=None):
    """OpticalSystem.seqtrace (raytracer/optical_system.py:73-94) for any duck-typed system"""
    ib = as_device_bundle(initialbundle, device)
    ib._ensure()
    (records, lengths) = flatten_sequence(system, elementsequence, ib.wave)
    if not records:
        raise UnsupportedError("empty sequence")
    crystals = sum(r["material"]["type"] != "isotropic" for r in records)
    absorbing = has_complex_eps(records)
    if (crystals or absorbing) and (splitup or crystals > MAX_FUSED_CRYSTALS or ib._dir is not None):
        if absorbing and splitup and crystals <= MAX_FUSED_CRYSTALS and ib._dir is None:
            return _seqtrace_fused_crystal(ib, records, lengths, split=True)
        if hasattr(system, "_seqtrace_generic"):        # (the plugin-granular loop of pyrate_amd's own classes)
            return system._seqtrace_generic(ib, elementsequence, splitup)
        if crystals <= MAX_FUSED_CRYSTALS and ib._dir is None:
            return _seqtrace_fused_crystal(ib, records, lengths, split=True)
        raise UnsupportedError("more than %d crystal interfaces (or a bundle with explicit directions) needs "
                               "pyrate_amd's material classes" % MAX_FUSED_CRYSTALS)
    return [seqtrace_fused(ib, records, lengths)]
