"""Python energy callables on the GPU: the reference's ``energy(real_params, complex_params) -> float`` (and its energy
dictionaries, and ``reject_condition``) TRACED into a HIP device function and compiled around the kernels as a user plugin.

The reference couples sampler and physics through a Python callable that it invokes once per proposal
(metropolis_engine.py:20, :111-120, :250).  A HIP kernel cannot call Python, and this engine has no CPU fallback -- so the
callable is executed ONCE, at construction, on symbolic stand-ins for the parameters: ``real_params`` / ``complex_params`` are
numpy object arrays of :class:`Sym` / :class:`SymComplex` nodes that overload arithmetic and record what the function
computes.  Whatever Python the function is written in -- lambdas, methods of a ``System`` object holding constants,
``*real_params`` unpacking, closures, helper functions, ``numpy`` ufuncs and reductions (the forms of README.md:26-33,
demo/toymodel_xypotentialwell.py:13-32 and demo/toymodel_complex_and_real.py:17-33) -- runs as ordinary Python; only
arithmetic on the parameters is recorded.  The recorded expression graph is written out as ``me_user_energy`` (straight-line
code, common subexpressions shared, sums of two products as explicit fused multiply-adds), compiled by hipcc
(``build.build_user_energy``) and loaded like any hand-written plugin (include/metropolis_user_energy.h).

What cannot be traced raises :class:`TraceError` at construction: control flow that depends on parameter VALUES
(``if x > 0``, ``max(x, y)``, ``math.exp(x)`` -- use ``np.exp`` --, ``float(x)``), because the trace sees one symbolic
evaluation, not one per chain.
"""
import hashlib
import math
import os

import numpy as np

from . import _capi
from .energy import EnergySpec, RejectSpec

_UNARY = {"neg": "-({0})", "abs": "me_py_abs({0})", "sqrt": "me_py_sqrt({0})", "exp": "me_py_exp({0})", "log": "me_py_log({0})",
          "sin": "me_py_sin({0})", "cos": "me_py_cos({0})", "tan": "me_py_tan({0})", "tanh": "me_py_tanh({0})",
          "sinh": "me_py_sinh({0})", "cosh": "me_py_cosh({0})", "arctan": "me_py_atan({0})"}
_BINARY = {"add": "({0} + {1})", "sub": "({0} - {1})", "mul": "({0} * {1})", "div": "({0} / {1})", "pow": "me_py_pow({0}, {1})",
           "arctan2": "me_py_atan2({0}, {1})"}
_COMPARE = {"lt": "<", "le": "<=", "gt": ">", "ge": ">=", "eq": "==", "ne": "!="}


class TraceError(TypeError):
    """The Python energy does something that depends on parameter values and cannot be recorded symbolically."""


class Sym:
    """One real-valued node of the recorded expression graph: ``op`` in {"x", "const", unary, binary} with ``args``."""
    __array_priority__ = 1000.0
    __slots__ = ("op", "args", "value")

    def __init__(self, op, args=(), value=None):
        self.op, self.args, self.value = op, tuple(args), value

    # ---- construction helpers
    @staticmethod
    def lift(v):
        if isinstance(v, Sym):
            return v
        if isinstance(v, (bool, np.bool_)):
            raise TraceError("a boolean cannot be used as a number in a traced energy")
        if isinstance(v, (int, float, np.integer, np.floating)):
            return Sym("const", value=float(v))
        if isinstance(v, (complex, np.complexfloating)) and complex(v).imag == 0.0:
            return Sym("const", value=float(complex(v).real))
        raise TraceError("cannot use %r (%s) in a traced energy" % (v, type(v).__name__))

    def _bin(self, op, other, swap=False):
        if isinstance(other, (SymComplex, complex, np.complexfloating)) and not (
                isinstance(other, (complex, np.complexfloating)) and complex(other).imag == 0.0):
            return NotImplemented if isinstance(other, SymComplex) else SymComplex(self, Sym.lift(0.0))._bin(op, other, swap)
        if isinstance(other, np.ndarray):
            return NotImplemented
        a, b = (Sym.lift(other), self) if swap else (self, Sym.lift(other))
        if a.op == "const" and b.op == "const":              # constant folding keeps the generated code short
            return Sym("const", value={"add": a.value + b.value, "sub": a.value - b.value, "mul": a.value * b.value,
                                       "div": a.value / b.value if b.value else math.nan}[op])
        # identities that complex arithmetic with real constants produces by the dozen (x * (1 + 0j) = x + 0 i, ...)
        zero = lambda n: n.op == "const" and n.value == 0.0
        one = lambda n: n.op == "const" and n.value == 1.0
        if op == "mul":
            if zero(a) or zero(b):
                return Sym("const", value=0.0)
            if one(a):
                return b
            if one(b):
                return a
        elif op == "add":
            if zero(a):
                return b
            if zero(b):
                return a
        elif op == "sub":
            if zero(b):
                return a
            if zero(a):
                return -b
        elif op == "div" and one(b):
            return a
        return Sym(op, (a, b))

    def __add__(self, o): return self._bin("add", o)
    def __radd__(self, o): return self._bin("add", o, True)
    def __sub__(self, o): return self._bin("sub", o)
    def __rsub__(self, o): return self._bin("sub", o, True)
    def __mul__(self, o): return self._bin("mul", o)
    def __rmul__(self, o): return self._bin("mul", o, True)
    def __truediv__(self, o): return self._bin("div", o)
    def __rtruediv__(self, o): return self._bin("div", o, True)
    def __neg__(self):
        if self.op == "const":
            return Sym("const", value=-self.value)
        return self.args[0] if self.op == "neg" else Sym("neg", (self,))

    def __pos__(self): return self
    def __abs__(self): return Sym("abs", (self,))

    def __pow__(self, o):
        if isinstance(o, (int, np.integer)) or (isinstance(o, (float, np.floating)) and float(o).is_integer() and abs(o) <= 64):
            n = int(o)
            if n == 0:
                return Sym.lift(1.0)
            out, base, k = None, self, abs(n)
            while k:                                           # square and multiply: x**4 is two multiplications
                if k & 1:
                    out = base if out is None else out * base
                k >>= 1
                if k:
                    base = base * base
            return out if n > 0 else 1.0 / out
        if isinstance(o, (float, np.floating)) and float(o) == 0.5:
            return Sym("sqrt", (self,))
        if isinstance(o, SymComplex):
            raise TraceError("complex exponents are not supported in a traced energy")
        return Sym("pow", (self, Sym.lift(o)))

    def __rpow__(self, o):
        return Sym("pow", (Sym.lift(o), self))

    # comparisons give boolean nodes (for reject_condition); using one in `if` / `and` / `or` is an error
    def _cmp(self, op, o): return SymBool("cmp", (self, Sym.lift(o)), op)
    def __lt__(self, o): return self._cmp("lt", o)
    def __le__(self, o): return self._cmp("le", o)
    def __gt__(self, o): return self._cmp("gt", o)
    def __ge__(self, o): return self._cmp("ge", o)
    def __eq__(self, o): return self._cmp("eq", o)
    def __ne__(self, o): return self._cmp("ne", o)
    __hash__ = object.__hash__

    def __float__(self):
        raise TraceError("the energy converts a parameter to a Python float (math.* functions, float(), int()): use the "
                         "numpy functions (np.exp, np.sqrt, ...) so that the computation can be recorded")
    __int__ = __index__ = __float__

    def __bool__(self):
        raise TraceError("the energy branches on a parameter value; a traced energy must be one arithmetic expression")

    # numpy object loops look these up by name (np.exp(x) -> x.exp())
    def conjugate(self): return self
    conj = conjugate
    real = property(lambda self: self)
    imag = property(lambda self: Sym.lift(0.0))
    def sqrt(self): return Sym("sqrt", (self,))
    def exp(self): return Sym("exp", (self,))
    def log(self): return Sym("log", (self,))
    def sin(self): return Sym("sin", (self,))
    def cos(self): return Sym("cos", (self,))
    def tan(self): return Sym("tan", (self,))
    def tanh(self): return Sym("tanh", (self,))
    def sinh(self): return Sym("sinh", (self,))
    def cosh(self): return Sym("cosh", (self,))
    def arctan(self): return Sym("arctan", (self,))
    def arctan2(self, o): return Sym("arctan2", (self, Sym.lift(o)))
    def square(self): return self * self
    def absolute(self): return abs(self)
    fabs = absolute


class SymBool:
    """A recorded comparison (or ``&`` / ``|`` / ``~`` of comparisons): the value of a traced ``reject_condition``."""
    __slots__ = ("op", "args", "cmp")

    def __init__(self, op, args, cmp=None):
        self.op, self.args, self.cmp = op, tuple(args), cmp

    def __and__(self, o): return SymBool("and", (self, SymBool.lift(o)))
    __rand__ = __and__
    def __or__(self, o): return SymBool("or", (self, SymBool.lift(o)))
    __ror__ = __or__
    def __invert__(self): return SymBool("not", (self,))

    @staticmethod
    def lift(v):
        if isinstance(v, SymBool):
            return v
        if isinstance(v, (bool, np.bool_)):
            return SymBool("const", (), bool(v))
        raise TraceError("cannot combine %r with a traced condition" % (v,))

    def __bool__(self):
        raise TraceError("a traced condition was used in `if` / `and` / `or` / `not`: combine comparisons with & | ~ instead "
                         "(the trace sees one symbolic evaluation, not one per chain)")


class SymComplex:
    """A complex parameter or intermediate as a pair of real nodes; complex arithmetic reduces to real arithmetic."""
    __array_priority__ = 1001.0
    __slots__ = ("re", "im")

    def __init__(self, re, im):
        self.re, self.im = Sym.lift(re), Sym.lift(im)

    @staticmethod
    def lift(v):
        if isinstance(v, SymComplex):
            return v
        if isinstance(v, Sym):
            return SymComplex(v, 0.0)
        if isinstance(v, (int, float, complex, np.number)) and not isinstance(v, (bool, np.bool_)):
            c = complex(v)
            return SymComplex(c.real, c.imag)
        raise TraceError("cannot use %r (%s) in a traced energy" % (v, type(v).__name__))

    def _bin(self, op, other, swap=False):
        if isinstance(other, np.ndarray):
            return NotImplemented
        a, b = (SymComplex.lift(other), self) if swap else (self, SymComplex.lift(other))
        if op == "add":
            return SymComplex(a.re + b.re, a.im + b.im)
        if op == "sub":
            return SymComplex(a.re - b.re, a.im - b.im)
        if op == "mul":
            return SymComplex(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re)
        if op == "div":
            den = b.re * b.re + b.im * b.im
            return SymComplex((a.re * b.re + a.im * b.im) / den, (a.im * b.re - a.re * b.im) / den)
        raise TraceError("unsupported complex operation %s" % op)

    def __add__(self, o): return self._bin("add", o)
    def __radd__(self, o): return self._bin("add", o, True)
    def __sub__(self, o): return self._bin("sub", o)
    def __rsub__(self, o): return self._bin("sub", o, True)
    def __mul__(self, o): return self._bin("mul", o)
    def __rmul__(self, o): return self._bin("mul", o, True)
    def __truediv__(self, o): return self._bin("div", o)
    def __rtruediv__(self, o): return self._bin("div", o, True)
    def __neg__(self): return SymComplex(-self.re, -self.im)
    def __pos__(self): return self
    def __abs__(self): return Sym("sqrt", (self.re * self.re + self.im * self.im,))

    def __pow__(self, o):
        if isinstance(o, (int, np.integer)) or (isinstance(o, (float, np.floating)) and float(o).is_integer() and abs(o) <= 64):
            n = int(o)
            if n == 0:
                return SymComplex(1.0, 0.0)
            out, base, k = None, self, abs(n)
            while k:
                if k & 1:
                    out = base if out is None else out * base
                k >>= 1
                if k:
                    base = base * base
            return out if n > 0 else 1.0 / out
        raise TraceError("only integer powers of complex values can be traced")

    def conjugate(self): return SymComplex(self.re, -self.im)
    conj = conjugate
    real = property(lambda self: self.re)
    imag = property(lambda self: self.im)
    def absolute(self): return abs(self)
    def square(self): return self * self

    def __bool__(self):
        raise TraceError("the energy branches on a parameter value; a traced energy must be one arithmetic expression")

    def __complex__(self):
        raise TraceError("the energy converts a parameter to a Python complex (cmath.* functions): use numpy functions")
    __float__ = __complex__


# --------------------------------------------------------------------------------------------------- tracing
def _inputs(n_real, n_complex):
    real = np.empty(n_real, dtype=object)
    for i in range(n_real):
        real[i] = Sym("x", value=i)
    cplx = np.empty(n_complex, dtype=object)
    for j in range(n_complex):
        cplx[j] = SymComplex(Sym("x", value=n_real + j), Sym("x", value=n_real + n_complex + j))
    return real, cplx


def _scalar(v):
    if isinstance(v, np.ndarray):
        if v.size != 1:
            raise TraceError("the energy must return one number, got an array of shape %s" % (v.shape,))
        v = v.reshape(-1)[0]
    return v


def trace_energy(fn, n_real, n_complex):
    """Run ``fn(real_params, complex_params)`` on symbolic parameters; returns the real :class:`Sym` it computes (the real
    part of a complex-typed result, SURVEY.md quirk Q11: the Landau toy returns ``complex128`` with zero imaginary part)."""
    real, cplx = _inputs(n_real, n_complex)
    out = _scalar(fn(real, cplx))
    if isinstance(out, SymComplex):
        out = out.re
    return Sym.lift(out)


def trace_reject(fn, n_real, n_complex):
    real, cplx = _inputs(n_real, n_complex)
    return SymBool.lift(_scalar(fn(real, cplx)))


# --------------------------------------------------------------------------------------------------- code generation
class _Emitter:
    """Straight-line C++ for a set of root nodes: every distinct (op, operands) is one ``const R tK``; ``a * b + c`` is
    written as one explicit fused multiply-add so that the value does not depend on which kernel the function is inlined in."""

    def __init__(self):
        self.lines, self.by_id, self.by_key, self.uses = [], {}, {}, {}

    @staticmethod
    def _leaf(n):
        return isinstance(n, Sym) and n.op in ("x", "const")

    def count(self, root):
        """How many times each node is used as an operand (a product used once may be folded into an fma)."""
        seen, stack = set(), [root]
        while stack:
            n = stack.pop()
            if id(n) in seen:
                continue
            seen.add(id(n))
            for a in n.args:
                self.uses[id(a)] = self.uses.get(id(a), 0) + 1
                stack.append(a)

    def operand(self, n):
        if isinstance(n, Sym) and n.op == "const":
            if math.isnan(n.value) or math.isinf(n.value):
                raise TraceError("the traced energy contains a non-finite constant")
            return "R(%s)" % repr(float(n.value))
        if isinstance(n, Sym) and n.op == "x":
            return "x[%d]" % n.value
        return self.by_id[id(n)]

    def emit(self, root):
        """Statements for everything ``root`` needs (iterative post-order: long sums are deep graphs); returns its name."""
        stack = [(root, False)]
        while stack:
            node, ready = stack.pop()
            if self._leaf(node) or id(node) in self.by_id:
                continue
            if not ready:
                stack.append((node, True))
                stack.extend((a, False) for a in node.args)
                continue
            args = [self.operand(a) for a in node.args]
            key = (type(node).__name__, node.op, getattr(node, "cmp", None)) + tuple(args)
            if key not in self.by_key:
                name = "%s%d" % ("b" if isinstance(node, SymBool) else "t", len(self.lines))
                self.lines.append(self._statement(node, name, args))
                self.by_key[key] = name
            self.by_id[id(node)] = self.by_key[key]
        return self.operand(root)

    def _statement(self, node, name, args):
        if isinstance(node, SymBool):
            if node.op == "cmp":
                expr = "(%s %s %s)" % (args[0], _COMPARE[node.cmp], args[1])
            elif node.op == "const":
                expr = "true" if node.cmp else "false"
            elif node.op == "not":
                expr = "(!%s)" % args[0]
            else:
                expr = "(%s %s %s)" % (args[0], "&&" if node.op == "and" else "||", args[1])
            return "const bool %s = %s;" % (name, expr)
        if node.op in ("add", "sub"):
            # a sum with a product in it: one explicit fma (the compiler would fuse one of the products anyway -- which one
            # may depend on the surrounding kernel)
            a, b = node.args
            if isinstance(b, Sym) and b.op == "mul" and self.uses.get(id(b), 0) <= 1:
                p, q = (self.operand(v) for v in b.args)
                return "const R %s = me_fma(%s%s, %s, %s);" % (name, "" if node.op == "add" else "-", p, q, args[0])
            if node.op == "add" and isinstance(a, Sym) and a.op == "mul" and self.uses.get(id(a), 0) <= 1:
                p, q = (self.operand(v) for v in a.args)
                return "const R %s = me_fma(%s, %s, %s);" % (name, p, q, args[1])
        template = _UNARY.get(node.op) or _BINARY.get(node.op)
        if template is None:
            raise TraceError("unsupported operation %s" % node.op)
        return "const R %s = %s;" % (name, template.format(*args))


_PRELUDE = '''// GENERATED by metropolisengine_amd.pyenergy from a Python energy callable -- do not edit.
#include "metropolis_user_energy.h"
__device__ __forceinline__ float me_py_abs(float v) { return __builtin_fabsf(v); }
__device__ __forceinline__ double me_py_abs(double v) { return __builtin_fabs(v); }
__device__ __forceinline__ float me_py_sqrt(float v) { return sqrtf(v); }
__device__ __forceinline__ double me_py_sqrt(double v) { return sqrt(v); }
__device__ __forceinline__ float me_py_exp(float v) { return expf(v); }
__device__ __forceinline__ double me_py_exp(double v) { return exp(v); }
__device__ __forceinline__ float me_py_log(float v) { return logf(v); }
__device__ __forceinline__ double me_py_log(double v) { return log(v); }
__device__ __forceinline__ float me_py_sin(float v) { return sinf(v); }
__device__ __forceinline__ double me_py_sin(double v) { return sin(v); }
__device__ __forceinline__ float me_py_cos(float v) { return cosf(v); }
__device__ __forceinline__ double me_py_cos(double v) { return cos(v); }
__device__ __forceinline__ float me_py_tan(float v) { return tanf(v); }
__device__ __forceinline__ double me_py_tan(double v) { return tan(v); }
__device__ __forceinline__ float me_py_tanh(float v) { return tanhf(v); }
__device__ __forceinline__ double me_py_tanh(double v) { return tanh(v); }
__device__ __forceinline__ float me_py_sinh(float v) { return sinhf(v); }
__device__ __forceinline__ double me_py_sinh(double v) { return sinh(v); }
__device__ __forceinline__ float me_py_cosh(float v) { return coshf(v); }
__device__ __forceinline__ double me_py_cosh(double v) { return cosh(v); }
__device__ __forceinline__ float me_py_atan(float v) { return atanf(v); }
__device__ __forceinline__ double me_py_atan(double v) { return atan(v); }
__device__ __forceinline__ float me_py_atan2(float a, float b) { return atan2f(a, b); }
__device__ __forceinline__ double me_py_atan2(double a, double b) { return atan2(a, b); }
__device__ __forceinline__ float me_py_pow(float a, float b) { return powf(a, b); }
__device__ __forceinline__ double me_py_pow(double a, double b) { return pow(a, b); }
'''


def _function(header, roots, ret):
    """One device function: the statements of all roots (shared subexpressions once), then ``ret(names)``."""
    em = _Emitter()
    for r in roots:
        em.count(r)
    names = [em.emit(r) for r in roots]
    body = "".join("  %s\n" % line for line in em.lines)
    return "%s {\n%s%s}\n" % (header, body, ret(names))


def generate_source(energy, n_real, n_complex, reject=None):
    """HIP source of the plugin for ``energy`` (a callable, or the reference's dictionary ``{"complex": {term: fn},
    "real": {...}, "all": {...}}``, metropolis_engine.py:111-116) and an optional ``reject`` callable.  Returns
    ``(source_text, term_names)``."""
    parts = [_PRELUDE]
    if isinstance(energy, dict):
        names = sorted(set().union(*[set(group) for group in energy.values()]))
        roots, groups = [], []
        for name in names:
            fn = None
            for group in ("all", "real", "complex"):
                fn = fn or energy.get(group, {}).get(name)
            roots.append(trace_energy(fn, n_real, n_complex))
            groups.append((1 if name in energy.get("real", {}) else 0) | (2 if name in energy.get("complex", {}) else 0))
        parts.append("#define ME_USER_N_TERMS %d\n" % len(names))
        parts.append("constexpr unsigned me_user_term_groups(int term) { return %s; }\n" % " : ".join(
            ["term == %d ? %du" % (t, g) for t, g in enumerate(groups[:-1])] + ["%du" % groups[-1]]))
        cases = []
        for t, root in enumerate(roots):
            cases.append(_function("  if (term == %d)" % t, [root], lambda n: "    return %s;\n  " % n[0]))
        parts.append("template <typename R>\n__device__ R me_user_energy_term(int term, const R *x, const R *coef) {\n%s  return R(0);\n}\n"
                     % "".join(cases))
        term_names = tuple(names)
    else:
        root = trace_energy(energy, n_real, n_complex)
        parts.append(_function("template <typename R>\n__device__ R me_user_energy(const R *x, const R *coef)", [root],
                               lambda n: "  return %s;\n" % n[0]))
        term_names = ("total",)
    if reject is not None:
        cond = trace_reject(reject, n_real, n_complex)
        parts.append("#define ME_USER_HAS_REJECT\n")
        parts.append(_function("template <typename R>\n__device__ bool me_user_reject(const R *x, const R *coef)", [cond],
                               lambda n: "  return %s;\n" % n[0]))
    return "".join(parts), term_names


class PythonEnergy(EnergySpec):
    """``energy_functions`` given as the reference gives it -- a Python callable ``(real_params, complex_params) -> float`` or
    its term dictionary -- traced at construction and compiled as a user plugin (see the module docstring).  ``reject`` is
    the reference's ``reject_condition`` callable (honoured when passed to the constructor; the reference drops it there,
    quirk Q6)."""
    kind = _capi.ENERGY_USER

    def __init__(self, energy, reject=None):
        if not (callable(energy) or isinstance(energy, dict)):
            raise TypeError("energy must be a callable or a dictionary of term callables")
        self.energy, self.reject = energy, reject
        self.term_names = tuple(sorted(set().union(*[set(g) for g in energy.values()]))) if isinstance(energy, dict) else ("total",)
        self.name = None
        self._loaded = set()

    def coefficients(self, n_real, n_complex):
        return np.zeros(0)                               # constants are literals of the generated source

    def build_plugin(self, n_real, n_complex):
        """Trace, write the generated source to ``_build/pyenergy/py<digest>.h`` and compile it (hipcc, once per distinct
        energy and dimensions; up to date plugins are reused).  Returns the plugin path; sets ``self.name``."""
        from . import build
        source, _ = generate_source(self.energy, n_real, n_complex, self.reject)
        digest = hashlib.sha256(("%d,%d\n" % (n_real, n_complex) + source).encode()).hexdigest()[:16]
        self.name = "py" + digest
        directory = os.path.join(build.OBJ_DIR, "pyenergy")
        os.makedirs(directory, exist_ok=True)
        path = os.path.join(directory, self.name + ".h")
        if not os.path.exists(path) or open(path).read() != source:
            with open(path, "w") as fh:
                fh.write(source)
        return build.build_user_energy(path, self.name, n_real, n_complex)

    def ensure_loaded(self, n_real, n_complex):
        if (n_real, n_complex) in self._loaded:
            return
        plugin = self.build_plugin(n_real, n_complex)
        _capi.check(_capi.load().me_load_plugin(plugin.encode()))
        self._loaded.add((n_real, n_complex))


class PythonReject(RejectSpec):
    """The traced ``reject_condition`` of a :class:`PythonEnergy` plugin (``ME_REJECT_USER``)."""
    kind = _capi.REJECT_USER
