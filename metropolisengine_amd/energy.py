"""Device-side energy and constraint specifications.

The reference couples the engine to the physics through a Python callable
``energy(real_params, complex_params) -> float`` (metropolis_engine.py:20, :111-120) and an optional
``reject_condition(real_params, complex_params) -> bool`` (:142-146).  A Python callable cannot run inside a HIP
kernel, so the GPU engine takes one of these specifications in the same positional slot.  Each maps to an
``me_energy_kind`` / ``me_reject_kind`` of the C ABI (include/metropolis_engine.h).
"""
import numpy as np

from . import _capi


class EnergySpec:
    """Base class: ``kind`` (me_energy_kind), ``coefficients(nr, nc)`` (flat float64 array) and ``term_names`` -- the
    keys of the reference's energy dictionary (metropolis_engine.py:111-118); a single function is ``("total",)``."""
    kind = None
    name = "energy"
    term_names = ("total",)

    def coefficients(self, n_real, n_complex):
        raise NotImplementedError


class IsoQuadratic(EnergySpec):
    """``E = a (sum x_i^2 + sum |z_j|^2)`` -- README.md:26-27 (a = 1, one real parameter)."""
    kind = _capi.ENERGY_ISO_QUAD

    def __init__(self, a=1.0):
        self.a = float(a)

    def coefficients(self, n_real, n_complex):
        return np.array([self.a], dtype=np.float64)


class DiagQuadratic(EnergySpec):
    """``E = sum a_i x_i^2 + sum b_j |z_j|^2`` (BASELINE config 3)."""
    kind = _capi.ENERGY_DIAG_QUAD

    def __init__(self, a=(), b=()):
        self.a = np.asarray(a, dtype=np.float64).ravel()
        self.b = np.asarray(b, dtype=np.float64).ravel()

    def coefficients(self, n_real, n_complex):
        if len(self.a) != n_real or len(self.b) != n_complex:
            raise ValueError("DiagQuadratic needs %d real and %d complex weights" % (n_real, n_complex))
        return np.concatenate((self.a, self.b))


class DenseQuadratic(EnergySpec):
    """``E = x^T A x`` over the real vector ``[real | Re z | Im z]`` (BASELINE config 4)."""
    kind = _capi.ENERGY_DENSE_QUAD

    def __init__(self, matrix):
        self.matrix = np.ascontiguousarray(matrix, dtype=np.float64)

    def coefficients(self, n_real, n_complex):
        d = n_real + 2 * n_complex
        if self.matrix.shape != (d, d):
            raise ValueError("DenseQuadratic needs a %dx%d matrix" % (d, d))
        return self.matrix.ravel()


class LandauToy(EnergySpec):
    """``k(1-x)^2 + k(1-y)^2 + x y (alpha |c|^2 + beta |c|^4)`` -- demo/toymodel_complex_and_real.py:17-29.

    ``terms=True`` is the form the demo actually passes (:31-33): the dictionary ``{"complex": {"field"}, "real":
    {"field", "area"}, "all": {"field", "area"}}``.  The engine then keeps one ledger row per term, and a group step
    re-evaluates and compares only its group's terms (metropolis_engine.py:214-221, :230-237)."""

    def __init__(self, k=1.0, alpha=-1.0, beta=0.5, terms=False):
        self.k, self.alpha, self.beta = float(k), float(alpha), float(beta)
        self.kind = _capi.ENERGY_LANDAU_TERMS if terms else _capi.ENERGY_LANDAU_TOY
        self.term_names = ("field", "area") if terms else ("total",)

    def coefficients(self, n_real, n_complex):
        if (n_real, n_complex) != (2, 1):
            raise ValueError("LandauToy is defined for 2 real + 1 complex parameters")
        return np.array([self.k, self.alpha, self.beta], dtype=np.float64)


class CylinderSurrogate(EnergySpec):
    """Cylinder-style surrogate energy (BASELINE config 5; definition in DESIGN.md and oracle/energies.py)."""
    kind = _capi.ENERGY_CYLINDER

    def __init__(self, kappa=1.0, gamma=0.5, wavenumber=1.0):
        self.kappa, self.gamma, self.wavenumber = float(kappa), float(gamma), float(wavenumber)

    def coefficients(self, n_real, n_complex):
        if n_real < 1 or n_complex < 1:
            raise ValueError("CylinderSurrogate needs at least one real and one complex parameter")
        return np.array([self.kappa, self.gamma, self.wavenumber], dtype=np.float64)


class UserEnergy(EnergySpec):
    """A user-written device function (include/metropolis_user_energy.h) -- the GPU form of the reference's
    ``energy(real_params, complex_params)`` callback (metropolis_engine.py:20).

    ``source`` is the HIP header defining ``me_user_energy``; it is compiled (hipcc, once per name and dimensions)
    around the engine's kernels into ``lib/libme_user_<name>_<nr>_<nc>.so`` and loaded with ``me_load_plugin``.
    ``indirect=True`` calls it through a ``__device__`` function pointer instead of inlining it.
    ``term_names``: for a term-wise plugin (``ME_USER_N_TERMS`` in the source) the names of its terms, in order.
    """

    def __init__(self, name, source=None, coefficients=(), indirect=False, term_names=("total",)):
        self.name = name
        self.term_names = tuple(term_names)
        self.source = source
        self.coeffs = np.asarray(coefficients, dtype=np.float64).ravel()
        self.kind = _capi.ENERGY_USER_INDIRECT if indirect else _capi.ENERGY_USER
        self._loaded = set()

    def coefficients(self, n_real, n_complex):
        return self.coeffs

    def ensure_loaded(self, n_real, n_complex):
        """Build (if a source is given and the plugin is stale) and load the plugin for these dimensions."""
        import os
        from . import build
        key = (n_real, n_complex)
        if key in self._loaded:
            return
        path = build.user_plugin_path(self.name, n_real, n_complex)
        if self.source is not None:
            path = build.build_user_energy(self.source, self.name, n_real, n_complex)
        elif not os.path.exists(path):
            raise _capi.MetropolisLibraryError("user-energy plugin %s not found and no source given" % path)
        _capi.check(_capi.load().me_load_plugin(path.encode()))
        self._loaded.add(key)


class RejectSpec:
    kind = _capi.REJECT_NONE
    bound = 0.0


class UserReject(RejectSpec):
    """The ``me_user_reject`` device function of the engine's :class:`UserEnergy` plugin (ME_REJECT_USER)."""
    kind = _capi.REJECT_USER


class AbsReal0AtLeast(RejectSpec):
    """Hard wall ``|x_0| >= bound`` (the legacy engine's ``abs(amplitude) >= 1``, /metropolis_engine.py:139-141)."""
    kind = _capi.REJECT_ABS_REAL0_GE

    def __init__(self, bound=1.0):
        self.bound = float(bound)
