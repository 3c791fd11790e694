"""
Scalar parameter holder with the accessors the reference's trace loop uses on its
``FloatOptimizableVariable`` (core/optimizable_variable.py:289-306): ``var()``,
``var.evaluate()``, ``var.set_value(v)``.  The optimiser state machine (fixed /
variable / pickup) is host bookkeeping outside the hot path.
"""
import uuid
import weakref

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
#   * ``FloatVariable.set_value`` (a variable belongs to EVERY object it was assigned to -- a pickup shared by two
#     surfaces invalidates both);
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
        value._add_owner(owner)
    elif type(value) is dict:
        value = TrackedDict(value, owner=owner)
    elif isinstance(value, TrackedDict):
        if value._owner is None:
            value._rehome(owner)
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

    def __reduce__(self):
        # copies and pickles come back as tracked dictionaries WITHOUT an owner; the object they end up in adopts them
        # (Named.__setstate__ -- copy.deepcopy and pickle restore an object's state without going through __setattr__)
        return (TrackedDict, (dict(self),))

    def _rehome(self, owner):
        """make ``owner`` the object whose epoch this dictionary's mutations (and those of the variables in it) move"""
        self._owner = owner
        for (k, v) in list(dict.items(self)):
            dict.__setitem__(self, k, _adopt(owner, v))


class FloatVariable(object):
    def __init__(self, value, name=""):
        self._value = _scalar(value)
        self.name = name
        self._owners = []            # weak references to the tracked objects the variable was assigned to

    def _add_owner(self, owner):
        if owner is None:
            return
        alive = [r for r in self._owners if r() is not None]
        if not any(r() is owner for r in alive):
            alive.append(weakref.ref(owner))
        self._owners = alive

    @property
    def _owner(self):
        """the object the variable was assigned to last (None: never assigned, or that object is gone)"""
        for r in reversed(self._owners):
            if r() is not None:
                return r()
        return None

    def __reduce__(self):            # a copy starts without owners: whoever it is put into adopts it
        return (FloatVariable, (self._value, self.name))

    def evaluate(self):
        return self._value

    def __call__(self):
        return self._value

    def set_value(self, value):
        self._value = _scalar(value)
        owners = [r() for r in self._owners]
        _touch(None)
        for o in owners:             # every holder's cached record is stale now, not just the last one's
            if o is not None:
                object.__setattr__(o, "_epoch", _TICK[0])

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

    def __setstate__(self, state):
        # copy.copy / copy.deepcopy / pickle hand the attribute dictionary over without calling __setattr__: adopt
        # every value here, so that the copy's dictionaries and variables move the COPY's epoch from now on
        # (a dictionary that already HAS an owner is somebody else's -- copy.copy passes the original's own attribute
        #  dictionary as the state --: the copy gets a tracked dictionary of its own with the same entries, the
        #  original keeps its one and its epoch)
        for (k, v) in state.items():
            if isinstance(v, TrackedDict):
                if v._owner is None:
                    v._rehome(self)
                elif v._owner is not self:
                    v = TrackedDict(dict(v), owner=self)
            object.__setattr__(self, k, _adopt(self, v))
        _touch(self)
