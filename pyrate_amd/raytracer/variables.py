"""
Scalar parameter holder with the accessors the reference's trace loop uses on its
``FloatOptimizableVariable`` (core/optimizable_variable.py:289-306): ``var()``,
``var.evaluate()``, ``var.set_value(v)``.  The optimiser state machine (fixed /
variable / pickup) is host bookkeeping outside the hot path.
"""
import uuid

import numpy as _np


def _scalar(value):
    """float, or complex when the value has an imaginary part (a complex refractive index: the reference's
    FixedState holds whatever it is given)"""
    if isinstance(value, complex) or getattr(value, "dtype", None) is not None and getattr(value.dtype, "kind", "") == "c":
        value = complex(value)
        return value if value.imag != 0.0 else float(value.real)
    return float(value)


# ---- mutation tracking -------------------------------------------------------------------------------------------
# The reference re-reads the whole object graph on every trace.  This package flattens the graph into a device
# table, and an optimiser loop (thousands of traces on small bundles) pays more for that walk than
# for the launch.  So every object of these mirror classes carries an EPOCH: the value of a process-wide counter at
# its last mutation.  Everything ``surface_table.flatten_sequence`` reads is reachable only through paths that
# advance the counter:
#   * attribute assignment on a Named object (``Named.__setattr__``) -- LocalCoordinates.update() re-assigns its
#     matrices, so a frame's epoch moves when its geometry does;
#   * ``FloatVariable.set_value`` (the variable belongs to the object it was assigned to);
#   * item assignment / deletion on the dictionaries these objects hold (``TrackedDict``: annotations, an element's
#     surfaces / materials, a shape's params);
#   * NumPy arrays held as attributes are stored as read-only copies: in-place mutation raises instead of going stale.
# ``mutation_epoch()`` unchanged = nothing changed = the flattened table of the last trace is still the table.
_TICK = [0]


def mutation_epoch():
    return _TICK[0]


def _touch(owner):
    _TICK[0] += 1
    if owner is not None:
        object.__setattr__(owner, "_epoch", _TICK[0])


def _adopt(owner, value):
    """what is stored when ``value`` is put into a tracked object or one of its dictionaries"""
    if isinstance(value, FloatVariable):
        value._owner = owner
    elif type(value) is dict:
        value = TrackedDict(value, owner=owner)
    elif isinstance(value, TrackedDict):
        if value._owner is None:
            value._owner = owner
    elif isinstance(value, _np.ndarray):
        value = _np.array(value)
        value.flags.writeable = False
    return value


class TrackedDict(dict):
    """a dict whose mutations count as mutations of the object that holds it"""

    def __init__(self, *args, owner=None, **kwargs):
        dict.__init__(self)
        self._owner = owner
        if args or kwargs:
            self.update(*args, **kwargs)

    def __setitem__(self, key, value):
        dict.__setitem__(self, key, _adopt(self._owner, value))
        _touch(self._owner)

    def __delitem__(self, key):
        dict.__delitem__(self, key)
        _touch(self._owner)

    def update(self, *args, **kwargs):
        for (k, v) in dict(*args, **kwargs).items():
            dict.__setitem__(self, k, _adopt(self._owner, v))
        _touch(self._owner)

    def setdefault(self, key, default=None):
        if key not in self:
            self[key] = default
        return dict.__getitem__(self, key)

    def pop(self, *args):
        out = dict.pop(self, *args)
        _touch(self._owner)
        return out

    def popitem(self):
        out = dict.popitem(self)
        _touch(self._owner)
        return out

    def clear(self):
        dict.clear(self)
        _touch(self._owner)

    def __ior__(self, other):
        self.update(other)
        return self

    def __reduce__(self):            # copies and pickles are plain, untracked dictionaries until somebody adopts them
        return (dict, (dict(self),))


class FloatVariable(object):
    def __init__(self, value, name=""):
        self._value = _scalar(value)
        self.name = name
        self._owner = None           # the tracked object the variable was assigned to

    def evaluate(self):
        return self._value

    def __call__(self):
        return self._value

    def set_value(self, value):
        self._value = _scalar(value)
        _touch(self._owner)

    def __repr__(self):
        return "FloatVariable(%r, name=%r)" % (self._value, self.name)


class Named(object):
    """name + kind + annotations, like core/base.py:31-89 gives every reference object; and the mutation tracking
    described above (``_epoch``)"""
    kind = "object"
    _epoch = 0

    def __init__(self, name=""):
        self.name = name if name != "" else str(uuid.uuid4())
        self.annotations = {}

    def __setattr__(self, key, value):
        object.__setattr__(self, key, _adopt(self, value))
        _touch(self)

    def set_name(self, name):
        self.name = name
