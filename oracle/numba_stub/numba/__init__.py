"""Identity stand-in for the `numba` package (build container only).

TEST INFRASTRUCTURE -- not product code.  The reference (neonnnnn/sparsepoly) is
pure Python decorated with ``@njit`` / ``@jitclass``; Numba itself is not
installable in this image.  Putting this directory ahead of ``/root/reference``
on ``PYTHONPATH`` makes the decorators no-ops, so the reference's *own source*
runs under CPython with the same float64 operations in the same order.  It is
used only by ``oracle/gen_golden.py`` to emit the fixtures in ``tests/golden``.
"""


def _identity_decorator(*args, **kwargs):
    # bare form: @njit
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    # parameterised form: @njit(cache=True) / @jitclass(spec)
    def wrap(obj):
        return obj

    return wrap


njit = _identity_decorator
jit = _identity_decorator


class _TypeToken(object):
    """Stands in for numba.float64 & co: supports tok[:], tok[:, :] and tok(x)."""

    def __init__(self, name, cast):
        self._name = name
        self._cast = cast

    def __getitem__(self, item):
        return self

    def __call__(self, *args):
        if len(args) == 1:
            return self._cast(args[0])
        return self

    def __repr__(self):
        return "numba_stub.%s" % self._name


float64 = _TypeToken("float64", float)
float32 = _TypeToken("float32", float)
int32 = _TypeToken("int32", int)
int64 = _TypeToken("int64", int)
boolean = _TypeToken("boolean", bool)
