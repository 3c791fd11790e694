"""Small LRU of device-resident surface tables keyed by their content, so that the
plugin-granular calls (Shape.intersect, Material.refract, ...) and repeated seqtrace
calls of an optimiser loop do not re-upload identical tables."""
from collections import OrderedDict

from .. import engine
from ..surface_table import surface_record, table_key

_CACHE = OrderedDict()
_MAX = 32
# the same record OBJECTS again = the same table (records are never modified once made; surface_table's memo hands
# out the same dictionaries for unchanged surfaces): a tuple of ids instead of a content key.  The records are kept,
# so their ids stay theirs.
_BY_IDENTITY = {}


def system_for(records, device):
    ident = (tuple(map(id, records)), device.index)
    hit = _BY_IDENTITY.get(ident)
    if hit is not None and hit[1]._h:
        return hit[1]
    key = (table_key(records), device.index)
    sysd = _CACHE.get(key)
    if sysd is None:
        sysd = engine.DeviceSystem(records, device.index)
        _CACHE[key] = sysd
        while len(_CACHE) > _MAX:
            (_, old) = _CACHE.popitem(last=False)
            old.close()
    else:
        _CACHE.move_to_end(key)
    if len(_BY_IDENTITY) > 256:
        _BY_IDENTITY.clear()
    _BY_IDENTITY[ident] = (list(records), sysd)
    return sysd


def clear():
    _BY_IDENTITY.clear()
    while _CACHE:
        (_, old) = _CACHE.popitem()
        old.close()


class _NoAperture(object):
    kind = "aperture"
    annotations = {}

    def __init__(self, lc):
        self.lc = lc


class _SurfaceView(object):
    """a (shape, aperture) pair seen as a surface by surface_record"""

    def __init__(self, shape, aperture):
        self.shape = shape
        self.aperture = aperture


def single_surface_system(shape, aperture, material, mirror, wave, device):
    """one-record table for plugin-granular calls; aperture None = no vignetting"""
    ap = aperture if aperture is not None else _NoAperture(shape.lc)
    rec = surface_record(_SurfaceView(shape, ap), material, mirror, wave)
    return system_for([rec], device)
