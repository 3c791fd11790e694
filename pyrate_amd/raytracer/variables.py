"""
Scalar parameter holder with the accessors the reference's trace loop uses on its
``FloatOptimizableVariable`` (core/optimizable_variable.py:289-306): ``var()``,
``var.evaluate()``, ``var.set_value(v)``.  The optimiser state machine (fixed /
variable / pickup) is host bookkeeping outside the hot path.
"""
import uuid


def _scalar(value):
    """float, or complex when the value has an imaginary part (a complex refractive index: the reference's
    FixedState holds whatever it is given)"""
    if isinstance(value, complex) or getattr(value, "dtype", None) is not None and getattr(value.dtype, "kind", "") == "c":
        value = complex(value)
        return value if value.imag != 0.0 else float(value.real)
    return float(value)


class FloatVariable(object):
    def __init__(self, value, name=""):
        self._value = _scalar(value)
        self.name = name

    def evaluate(self):
        return self._value

    def __call__(self):
        return self._value

    def set_value(self, value):
        self._value = _scalar(value)

    def __repr__(self):
        return "FloatVariable(%r, name=%r)" % (self._value, self.name)


class Named(object):
    """name + kind + annotations, like core/base.py:31-89 gives every reference object"""
    kind = "object"

    def __init__(self, name=""):
        self.name = name if name != "" else str(uuid.uuid4())
        self.annotations = {}

    def set_name(self, name):
        self.name = name
