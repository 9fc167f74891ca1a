"""Import harness for the *reference* (michchr/pyhybridcontrol at /root/reference).

TEST INFRASTRUCTURE ONLY -- used by oracle/gen_golden.py, in the build container, to run the
reference's own numpy code (block_toeplitz / MldModel / MldEvoMatrices / ObjectiveAtoms weights)
and dump golden vectors into tests/golden/.  Nothing in the product path imports this file, and it
is a no-op (raises ReferenceUnavailable) when /root/reference is absent (e.g. on the GPU box).

What is stubbed and why:
  * `wrapt` is not installed: a minimal decorator library stand-in (signature-preserving function
    wrappers with descriptor binding).  It carries no numerics.
  * `cvxpy` is not installed: an inert stand-in exposing only the *names* the reference touches at
    import time (Expression, Parameter, Variable, error.SolverError).  No solver behaviour is
    emulated; every reference code path that needs a real cvxpy object is out of reach and is NOT
    used for fixtures.
  * numpy/collections names removed since ~2019 are aliased (np.NaN, np.int, np.str,
    collections.Container/Sequence).
The reference's own files are imported from where they lie; nothing is copied.
"""
import collections
import collections.abc
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("PYHC_REFERENCE_ROOT", "/root/reference")


class ReferenceUnavailable(RuntimeError):
    pass


# --------------------------------------------------------------------------- wrapt stand-in
def _make_wrapt():
    wrapt = types.ModuleType("wrapt")
    decorators = types.ModuleType("wrapt.decorators")

    class _BoundWrapper(object):
        def __init__(self, parent, instance, bound_wrapped):
            self._self_parent = parent
            self._self_instance = instance
            self.__wrapped__ = bound_wrapped

        def __call__(self, *args, **kwargs):
            p = self._self_parent
            return p._self_wrapper(self.__wrapped__, self._self_instance, args, kwargs)

        def __getattr__(self, name):
            return getattr(self.__wrapped__, name)

    class FunctionWrapper(object):
        def __init__(self, wrapped, wrapper, enabled=None, adapter=None):
            object.__setattr__(self, "__wrapped__", wrapped)
            object.__setattr__(self, "_self_wrapper", wrapper)
            object.__setattr__(self, "_self_enabled", enabled)
            object.__setattr__(self, "_self_adapter", adapter)
            for attr in ("__name__", "__qualname__", "__doc__", "__module__"):
                try:
                    object.__setattr__(self, attr, getattr(wrapped, attr))
                except (AttributeError, TypeError):
                    pass

        def __get__(self, instance, owner):
            if instance is None:
                return self
            bound = self.__wrapped__.__get__(instance, owner)
            return _BoundWrapper(self, instance, bound)

        def __call__(self, *args, **kwargs):
            return self._self_wrapper(self.__wrapped__, None, args, kwargs)

        def __getattr__(self, name):
            # only reached when normal lookup fails
            return getattr(object.__getattribute__(self, "__wrapped__"), name)

        def __setattr__(self, name, value):
            if name.startswith("_self_") or name in ("__wrapped__", "__name__", "__qualname__", "__doc__"):
                object.__setattr__(self, name, value)
            else:
                setattr(self.__wrapped__, name, value)

        @property
        def __signature__(self):
            import inspect
            target = self._self_adapter if self._self_adapter is not None else self.__wrapped__
            return inspect.signature(target)

    class AdapterWrapper(FunctionWrapper):
        def __init__(self, *args, **kwargs):
            adapter = kwargs.pop("adapter", None)
            super(AdapterWrapper, self).__init__(*args, **kwargs)
            object.__setattr__(self, "_self_adapter", adapter)

    def decorator(wrapper=None, enabled=None, adapter=None):
        if wrapper is None:
            def _partial(w):
                return decorator(w, enabled=enabled, adapter=adapter)
            return _partial

        def _apply(wrapped):
            return AdapterWrapper(wrapped=wrapped, wrapper=_as_wrapper(wrapper), enabled=enabled, adapter=adapter)

        # a decorator declared inside a class body / used on methods: wrapper(wrapped, instance, args, kwargs)
        return _DecoratorObject(wrapper, _apply)

    def _as_wrapper(wrapper):
        return wrapper

    class _DecoratorObject(object):
        """Result of @wrapt.decorator: callable on the function to decorate; also a descriptor so a
        decorator *defined as a method* (CallableMatrix._matrix_wrapper style) keeps working."""

        def __init__(self, wrapper, apply):
            self._wrapper = wrapper
            self._apply = apply
            self.__wrapped__ = wrapper

        def __call__(self, wrapped):
            return self._apply(wrapped)

    wrapt.decorator = decorator
    wrapt.FunctionWrapper = FunctionWrapper
    wrapt.ObjectProxy = FunctionWrapper
    decorators.AdapterWrapper = AdapterWrapper
    wrapt.decorators = decorators
    return wrapt, decorators


# --------------------------------------------------------------------------- cvxpy stand-in
def _make_cvxpy():
    cvx = types.ModuleType("cvxpy")
    expressions = types.ModuleType("cvxpy.expressions")
    expression = types.ModuleType("cvxpy.expressions.expression")
    error = types.ModuleType("cvxpy.error")

    class Expression(object):
        pass

    class _Inert(Expression):
        def __init__(self, *a, **k):
            raise NotImplementedError("cvxpy is not installed; reference solver paths are out of reach")

    class SolverError(Exception):
        pass

    expression.Expression = Expression
    expressions.expression = expression
    error.SolverError = SolverError
    cvx.expressions = expressions
    cvx.error = error
    cvx.Parameter = _Inert
    cvx.Variable = _Inert
    cvx.Problem = _Inert
    cvx.Constant = _Inert
    cvx.GUROBI = "GUROBI"
    cvx.CPLEX = "CPLEX"
    return cvx, expressions, expression, error


_installed = False


def install():
    """Make `import utils.matrix_utils`, `import models.mld_model`, ... resolve to the reference."""
    global _installed
    if _installed:
        return
    if not os.path.isdir(REFERENCE_ROOT):
        raise ReferenceUnavailable("reference tree not present at %s" % REFERENCE_ROOT)
    import numpy as np

    for name, val in (("NaN", np.nan), ("int", int), ("str", str), ("bool", bool), ("float", float)):
        if name not in np.__dict__:
            setattr(np, name, val)
    for name in ("Container", "Sequence", "Mapping", "MutableMapping", "Iterable", "Callable", "Hashable"):
        if not hasattr(collections, name):
            setattr(collections, name, getattr(collections.abc, name))

    if "wrapt" not in sys.modules:
        wrapt, decorators = _make_wrapt()
        sys.modules["wrapt"] = wrapt
        sys.modules["wrapt.decorators"] = decorators
    if "cvxpy" not in sys.modules:
        cvx, expressions, expression, error = _make_cvxpy()
        sys.modules["cvxpy"] = cvx
        sys.modules["cvxpy.expressions"] = expressions
        sys.modules["cvxpy.expressions.expression"] = expression
        sys.modules["cvxpy.error"] = error
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _installed = True


def uninstall():
    """Remove the reference from sys.path / sys.modules again (keeps test processes clean)."""
    global _installed
    if REFERENCE_ROOT in sys.path:
        sys.path.remove(REFERENCE_ROOT)
    for mod in list(sys.modules):
        root = mod.split(".")[0]
        if root in ("utils", "models", "controllers", "structdict", "examples", "tools", "wrapt", "cvxpy"):
            m = sys.modules[mod]
            f = getattr(m, "__file__", None)
            if f is None or f.startswith(REFERENCE_ROOT):
                del sys.modules[mod]
    _installed = False
