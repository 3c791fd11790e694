"""
pyrate_amd -- MI355X (gfx950) sequential raytrace engine behind mess42/pyrate's
RayBundle / Surface.intersect / Material.refract API.

    pyrate_amd.csrc/        HIP kernels + C ABI (libprt.so, include/prt.h)
    pyrate_amd._lib         ctypes binding (fails loudly when libprt.so is missing)
    pyrate_amd.engine       device-tensor level API (DeviceSystem.trace ...)
    pyrate_amd.surface_table  object graph -> POD surface table
    pyrate_amd.raytracer    host-side mirror of the reference classes (drop-in)
    pyrate_amd.systems      BASELINE workloads
    pyrate_amd.distributed  ray sharding over ranks + image-plane all-gather
"""
__version__ = "0.1.0"


def __getattr__(name):
    """lazy re-exports of the convenience entry points (pyrateoptics/__init__.py:83-465)"""
    if name in ("build_rotationally_symmetric_optical_system", "build_simple_optical_element",
                "build_simple_optical_system", "raytrace"):
        from . import builders
        return getattr(builders, name)
    if name in ("GlassCatalog", "CatalogMaterial"):
        from .raytracer.material import material_glasscat
        return getattr(material_glasscat, name)
    raise AttributeError(name)
