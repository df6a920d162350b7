"""numba.experimental stand-in: jitclass(spec) returns the class unchanged."""
from numba import _identity_decorator

jitclass = _identity_decorator
