"""ctypes binding of oracle/c/libme_oracle.so -- TEST / BENCH INFRASTRUCTURE ONLY (see oracle/c/me_oracle.c)."""
import ctypes
import os
import subprocess

import numpy as np

from .reference_chain import adaptation_constants

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c")
_LIB = os.path.join(_DIR, "libme_oracle.so")
_dp = ctypes.POINTER(ctypes.c_double)


def load(build=True):
    if not os.path.exists(_LIB):
        if not build:
            raise OSError("oracle/c/libme_oracle.so missing: run `make -C oracle/c`")
        subprocess.run(["make", "-C", _DIR, "-s", "libme_oracle.so"], check=True)
    lib = ctypes.CDLL(_LIB)
    lib.meo_create.restype = ctypes.c_void_p
    lib.meo_create.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint64,
                               ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _dp, _dp]
    lib.meo_destroy.argtypes = [ctypes.c_void_p]
    lib.meo_step.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    for name in ("meo_params", "meo_widths", "meo_energies"):
        getattr(lib, name).restype = _dp
        getattr(lib, name).argtypes = [ctypes.c_void_p]
    for name in ("meo_accepted", "meo_proposed"):
        getattr(lib, name).restype = ctypes.c_int64
        getattr(lib, name).argtypes = [ctypes.c_void_p]
    return lib


class COracle:
    """N chains, identity proposal shape, diagonal quadratic energy ``sum a x^2 + sum b |z|^2``, step_all only."""

    def __init__(self, nr, nc, a=(), b=(), n_chains=1, seed=0, temp=0.0, initial_real_params=None,
                 initial_complex_params=None, sampling_width=0.05, target_acceptance=0.3, chain_offset=0):
        self.lib = load()
        self.nr, self.nc, self.dim, self.n_chains = nr, nc, nr + 2 * nc, n_chains
        weights = np.ascontiguousarray(np.concatenate((np.asarray(a, float), np.asarray(b, float), np.asarray(b, float))))
        x0 = np.zeros(self.dim)
        if nr:
            x0[:nr] = np.asarray(initial_real_params, dtype=np.float64)
        if nc:
            z0 = np.asarray(initial_complex_params, dtype=np.complex128)
            x0[nr:nr + nc], x0[nr + nc:] = z0.real, z0.imag
        _, _, ratio = adaptation_constants(nr, nc, target_acceptance)
        self.handle = ctypes.c_void_p(self.lib.meo_create(nr, nc, n_chains, seed, chain_offset, temp, target_acceptance,
                                                          sampling_width, ratio, weights.ctypes.data_as(_dp),
                                                          x0.ctypes.data_as(_dp)))

    def step(self, n_sweeps=1):
        self.lib.meo_step(self.handle, n_sweeps)

    def _view(self, fn, shape):
        return np.ctypeslib.as_array(fn(self.handle), shape=shape).copy()

    @property
    def x(self):
        return self._view(self.lib.meo_params, (self.n_chains, self.dim))

    @property
    def width(self):
        return self._view(self.lib.meo_widths, (self.n_chains,))

    @property
    def energy(self):
        return self._view(self.lib.meo_energies, (self.n_chains,))

    @property
    def accepted(self):
        return self.lib.meo_accepted(self.handle)

    @property
    def proposed(self):
        return self.lib.meo_proposed(self.handle)

    def close(self):
        if self.handle:
            self.lib.meo_destroy(self.handle)
            self.handle = None

    def __del__(self):
        self.close()
