"""Small LRU of device-resident surface tables keyed by their content, so that the
plugin-granular calls (Shape.intersect, Material.refract, ...) and repeated seqtrace
calls of an optimiser loop do not re-upload identical tables."""
from collections import OrderedDict

from .. import engine
from ..surface_table import surface_record, table_key, _MEMO_RECORD_IDS

_CACHE = OrderedDict()
_MAX = 32
# the same record OBJECTS again = the same table (records are never modified once made; surface_table's memo hands
# out the same dictionaries for unchanged surfaces): a tuple of ids instead of a content key.  The records are kept,
# so their ids stay theirs.
_BY_IDENTITY = {}


def system_for(records, device):
    # the very list OpticalSystem._flattened handed out last time (nothing changed since): its device system hangs on
    # it -- valid while that system has not been overwritten for another table (DeviceSystem.update counts)
    pinned = getattr(records, "device_systems", None)
    if pinned is not None:
        ent = pinned.get(device.index)
        if ent is not None and ent[0]._h and ent[0].updates == ent[1]:
            return _on_stream(ent[0], device)
    # (identity stands for content only for records surface_table's memo made and keeps -- nobody else holds those
    #  dictionaries to edit them; a table a caller built by hand may have been edited in place: keyed by content)
    owned = all(id(r) in _MEMO_RECORD_IDS for r in records)
    ident = (tuple(map(id, records)), device.index)
    hit = _BY_IDENTITY.get(ident) if owned else None
    if hit is not None and hit[1]._h:
        if pinned is not None:
            pinned[device.index] = (hit[1], hit[1].updates)
        return _on_stream(hit[1], device)
    key = (table_key(records), device.index)
    sysd = _CACHE.get(key)
    if sysd is None:
        sysd = _recycled(records, device, key)
        if sysd is None:
            sysd = engine.DeviceSystem(records, device.index)
        _CACHE[key] = sysd
        while len(_CACHE) > _MAX:
            (_, old) = _CACHE.popitem(last=False)
            old.close()
    else:
        _CACHE.move_to_end(key)
    if owned:
        if len(_BY_IDENTITY) > 256:
            _BY_IDENTITY.clear()
        _BY_IDENTITY[ident] = (list(records), sysd)
    if pinned is not None:
        pinned[device.index] = (sysd, sysd.updates)
    return _on_stream(sysd, device)


def _on_stream(sysd, device):
    """remember the stream this system's launches go out on (the caller launches on the current one); a system that
    has been used on more than one stream is never overwritten in place (_recycled)"""
    cur = engine.raw_stream(device)
    seen = getattr(sysd, "_dispatch_stream", None)
    if seen is None:
        sysd._dispatch_stream = cur
    elif seen != cur:
        sysd._dispatch_stream = -1
    return sysd


# A table never seen before is the NORMAL case of an optimiser loop (every merit evaluation moves a parameter).  Rather
# than building a new device system for it -- three allocations, three blocking copies, and sooner or later three
# frees for the one that falls out of the cache --, the least recently used cached system of the same size is
# overwritten in place (DeviceSystem.update: one asynchronous copy, ordered on the current stream behind the launches
# that still use the old content).  Only once the cache holds sixteen tables (wavelength / field sweeps that alternate
# between a handful of tables keep them all), and only if the old system's launches went out on the stream that is
# current now (the copy is ordered against that stream only).
_RECYCLE_FROM = 16


def _recycled(records, device, key):
    if len(_CACHE) < _RECYCLE_FROM:
        return None
    stream = engine.raw_stream(device)
    for (old_key, cand) in _CACHE.items():                # oldest first
        if old_key[1] != device.index or cand.n_surfaces != len(records) or not cand._h:
            continue
        if getattr(cand, "_dispatch_stream", None) != stream:
            continue
        try:
            took = cand.update(records)
        except Exception:
            # the update failed -- on the device (the system is poisoned: engine.DeviceSystem.update closed it) or
            # before anything reached it (packing the records): either way the candidate leaves the caches CLOSED,
            # so that its device arrays do not leak
            cand.close()
            del _CACHE[old_key]
            for (ident, (_, s)) in list(_BY_IDENTITY.items()):
                if s is cand:
                    del _BY_IDENTITY[ident]
            raise
        if took:
            del _CACHE[old_key]
            for (ident, (_, s)) in list(_BY_IDENTITY.items()):
                if s is cand:
                    del _BY_IDENTITY[ident]
            return cand
        return None
    return None


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
