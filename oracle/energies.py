"""Vectorised (over chains) energy functions for the many-chain oracle -- TEST INFRASTRUCTURE ONLY.

State layout: ``x[n_chains, D]`` float64 with ``D = nr + 2 nc``: the real parameters, then the real parts, then
the imaginary parts of the complex parameters.  Each factory returns ``energy(x) -> [n_chains]``; a matching
``as_reference_callable`` gives the ``energy(real_params, complex_params) -> float`` form the reference takes
(metropolis_engine.py:20), so the same workload can be run through ``oracle.reference_chain``.
"""
import numpy as np


def split(x, nr, nc):
    return x[:, :nr], x[:, nr:nr + nc], x[:, nr + nc:nr + 2 * nc]


def iso_quadratic(nr, nc, a=1.0):
    """``E = a (sum x_i^2 + sum |z_j|^2)`` -- README.md:26-27, demo/toymodel_xypotentialwell.py:13-18."""
    return lambda x: a * np.sum(x * x, axis=1)


def diag_quadratic(nr, nc, a=(), b=()):
    """``E = sum a_i x_i^2 + sum b_j |z_j|^2`` (BASELINE config 3)."""
    w = np.concatenate((np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64),
                        np.asarray(b, dtype=np.float64)))
    assert w.shape[0] == nr + 2 * nc
    return lambda x: np.sum(w * x * x, axis=1)


def dense_quadratic(nr, nc, matrix):
    """``E = x^T A x`` over the real D-vector (BASELINE config 4)."""
    a = np.asarray(matrix, dtype=np.float64)
    assert a.shape == (nr + 2 * nc,) * 2
    return lambda x: np.einsum("ni,ij,nj->n", x, a, x)


def landau_toy(k=1.0, alpha=-1.0, beta=0.5):
    """``k(1-x)^2 + k(1-y)^2 + x y (alpha |c|^2 + beta |c|^4)`` -- demo/toymodel_complex_and_real.py:17-26."""
    def energy(x):
        xx, yy = x[:, 0], x[:, 1]
        a2 = x[:, 2] ** 2 + x[:, 3] ** 2
        return k * (1 - xx) ** 2 + k * (1 - yy) ** 2 + xx * yy * (alpha * a2 + beta * a2 * a2)
    return energy


def cylinder_surrogate(nr, nc, kappa=1.0, gamma=0.5, wavenumber=1.0):
    """Cylinder-style surrogate for BASELINE config 5 (the real ``cylinder`` energy is not available offline).

    Parameters: ``x0`` = surface amplitude (hard wall ``|x0| >= 1`` handled by the reject predicate, after
    /metropolis_engine.py:139-141), ``x1..`` = further real shape parameters, ``z_j`` = Fourier coefficients of a
    complex field with mode number ``q_j = j - (nc-1)/2``.  Surface term ``kappa (x0^2 + sum x_i^2)/(1 - x0^2)``
    (stiffening towards the wall), field term ``sum_j (gamma + (wavenumber q_j)^2 (1 + x0^2/2)) |z_j|^2
    + 0.5 (sum_j |z_j|^2)^2`` (amplitude-coupled gradient energy plus a quartic Landau term).
    """
    q = np.arange(nc, dtype=np.float64) - (nc - 1) / 2.0

    def energy(x):
        xr, zre, zim = split(x, nr, nc)
        x0 = xr[:, 0]
        surface = kappa * np.sum(xr * xr, axis=1) / (1.0 - x0 * x0)
        mod2 = zre * zre + zim * zim
        stiff = gamma + (wavenumber * q) ** 2 * (1.0 + 0.5 * x0 * x0)[:, None]
        tot = np.sum(mod2, axis=1)
        return surface + np.sum(stiff * mod2, axis=1) + 0.5 * tot * tot
    return energy


def wall_reject(bound=1.0):
    """Reject predicate ``|x0| >= bound`` evaluated before the energy (metropolis_engine.py:247-249)."""
    return lambda x: np.abs(x[:, 0]) >= bound


def as_reference_callable(energy, nr, nc):
    """Wrap a vectorised energy as the reference's ``energy(real_params, complex_params) -> float``."""
    def fn(real_params, complex_params):
        r = np.zeros(0) if real_params is None else np.asarray(real_params, dtype=np.float64)
        c = np.zeros(0, dtype=np.complex128) if complex_params is None else np.asarray(complex_params,
                                                                                          dtype=np.complex128)
        x = np.concatenate((r, c.real, c.imag))[None, :]
        return float(energy(x)[0])
    return fn


def as_reference_reject(reject, nr, nc):
    def fn(real_params, complex_params):
        r = np.zeros(0) if real_params is None else np.asarray(real_params, dtype=np.float64)
        c = np.asarray(complex_params, dtype=np.complex128) if complex_params is not None and len(complex_params) \
            else np.zeros(0, dtype=np.complex128)
        x = np.concatenate((r, c.real, c.imag))[None, :]
        return bool(reject(x)[0])
    return fn
