from . import raster  # noqa: F401
