"""Many-chain CPU restatement (numpy, float64) -- TEST INFRASTRUCTURE ONLY.

N independent copies of the reference chain (metropolis_engine.py:17-463, restated literally in
``oracle.reference_chain``), vectorised over chains and driven by the counter-based Philox streams of
``oracle.philox``.  This is the exact semantics the HIP kernels implement:

 * proposal   ``x' = x + sigma_r L_r g_r``,  ``z' = z + sigma_c L_c w``,  ``w = (g_re + i g_im)/sqrt 2``,
              ``L_r = chol(C_r)``, ``L_c = chol(conj(C_c))``          (metropolis_engine.py:261-302, quirk Q3)
 * accept     ``dE <= 0`` or (``T > 0`` and ``u <= exp(-dE/T)``)        (:319-338)
 * width      Robbins-Monro with the reference's ``ratio``              (:429-456)
 * measure    running mean / Haario covariance with the undivided epsilon term / observables, the
              covariance (and hence the factors) only once ``measure_step_counter > 50``    (:342-427)

``tests/test_oracle_manychain.py`` checks chain ``c`` of this class against ``ReferenceChain`` fed the same
Philox words, which ties it to the golden-pinned restatement.
"""
import math

import numpy as np

from . import philox
from .reference_chain import ADAPTATION_FLOOR, MEASURES_BEFORE_COVARIANCE, adaptation_constants


class ManyChainOracle:
    def __init__(self, nr, nc, energy, n_chains, seed=0, temp=0.0, initial_real_params=None,
                 initial_complex_params=None, sampling_width=0.05, target_acceptance=0.3, chain_offset=0,
                 reject=None, covariance_matrix_real=None, covariance_matrix_complex=None, adapt_shape=True):
        if nr + nc == 0:
            raise ValueError("need at least one real or complex parameter")
        assert temp is not None and temp >= 0
        self.nr, self.nc, self.dim = nr, nc, nr + 2 * nc
        self.n_chains = n_chains
        self.adapt_shape = adapt_shape     # False: the proposal shape stays the initial matrix (engine cov_mode="fixed")
        self.energy_fn = energy
        self.reject_fn = reject
        self.seed = seed
        self.temp = float(temp)
        self.chain_ids = np.arange(n_chains, dtype=np.uint64) + np.uint64(chain_offset)
        self.mode = "mixed" if (nr and nc) else ("real" if nr else "complex")
        self.target_acceptance = target_acceptance
        self.alpha, self.m, self.ratio = adaptation_constants(nr, nc, target_acceptance)

        x0 = np.zeros(self.dim)
        if nr:
            x0[:nr] = np.asarray(initial_real_params, dtype=np.float64)
        if nc:
            z0 = np.asarray(initial_complex_params, dtype=np.complex128)
            x0[nr:nr + nc], x0[nr + nc:] = z0.real, z0.imag
        self.x = np.broadcast_to(x0, (n_chains, self.dim)).copy()   # every chain starts where the reference's one chain does
        self.energy = np.asarray(energy(self.x), dtype=np.float64).copy()
        self.width_real = np.full(n_chains, float(sampling_width))
        self.width_complex = np.full(n_chains, float(sampling_width))
        self.width_all = np.full(n_chains, float(sampling_width))      # mixed engines' shared sampling_width
        self.step_index = 0
        self.measure_step_counter = 1
        self.accepted = 0
        self.proposed = 0
        self.last_accept = np.zeros(n_chains, dtype=bool)

        self.mean = self.x.copy()
        c_r = np.identity(nr) if covariance_matrix_real is None else np.asarray(covariance_matrix_real, np.float64)
        c_c = (np.identity(nc, dtype=np.complex128) if covariance_matrix_complex is None
               else np.asarray(covariance_matrix_complex, np.complex128))
        self.cov_real = np.broadcast_to(c_r, (n_chains, nr, nr)).copy()
        self.cov_complex = np.broadcast_to(c_c, (n_chains, nc, nc)).copy()
        self.observables_mean = self.observables()
        self._refresh_factors()

    # ------------------------------------------------------------------ helpers
    def complex_params(self):
        return self.x[:, self.nr:self.nr + self.nc] + 1j * self.x[:, self.nr + self.nc:]

    def observables(self):
        xr = self.x[:, :self.nr]
        return np.concatenate((np.abs(xr), np.abs(self.complex_params()), xr * xr), axis=1)

    def _refresh_factors(self):
        self.factor_real = np.linalg.cholesky(self.cov_real) if self.nr else np.zeros((self.n_chains, 0, 0))
        self.factor_complex = (np.linalg.cholesky(np.conj(self.cov_complex)) if self.nc
                               else np.zeros((self.n_chains, 0, 0), dtype=np.complex128))

    # ------------------------------------------------------------------ step
    def step(self, n_sweeps=1, group="all"):
        """``group`` = "all" (step_all, :241-259), "real" / "complex" (step_*_group called directly, :209-239)."""
        for _ in range(n_sweeps):
            self._sweep(group)

    def _decide(self, prop, u):
        """Wall, energy, accept rule (:247-252, :319-338) and commit for a proposed state; returns the accept mask."""
        rejected = self.reject_fn(prop) if self.reject_fn is not None else np.zeros(self.n_chains, dtype=bool)
        new_energy = np.asarray(self.energy_fn(prop), dtype=np.float64)
        diff = new_energy - self.energy
        if self.temp > 0:
            with np.errstate(over="ignore", invalid="ignore"):
                uphill_ok = u <= np.exp(-diff / self.temp)
        else:
            uphill_ok = np.zeros(self.n_chains, dtype=bool)
        accept = ~rejected & ((diff <= 0) | uphill_ok)
        self.x[accept] = prop[accept]
        self.energy[accept] = new_energy[accept]
        self.last_accept = accept
        self.accepted += int(np.sum(accept))
        self.proposed += self.n_chains
        return accept

    def _adapt(self, width, accept):
        damping = max(self.measure_step_counter / self.m, ADAPTATION_FLOOR)
        p = self.target_acceptance
        scale = width * self.ratio
        return np.where(accept, width + scale * (1 - p) / damping, width - scale * p / damping)

    def _sweep(self, group="all"):
        nr, nc = self.nr, self.nc
        # the word layout of a step does not depend on which group moves: normal i belongs to coordinate i
        g, u = philox.step_draws(self.seed, self.chain_ids, self.step_index, self.dim)
        prop = self.x.copy()
        if nr and group in ("all", "real"):
            prop[:, :nr] += self.width_real[:, None] * np.einsum("nij,nj->ni", self.factor_real, g[:, :nr])
        if nc and group in ("all", "complex"):
            w = (g[:, nr:nr + nc] + 1j * g[:, nr + nc:]) / math.sqrt(2.0)
            dz = self.width_complex[:, None] * np.einsum("nij,nj->ni", self.factor_complex, w)
            prop[:, nr:nr + nc] += dz.real
            prop[:, nr + nc:] += dz.imag
        accept = self._decide(prop, u)
        if group == "all" and self.mode == "mixed":
            # one shared width is adapted and mirrored into both group widths (:429-438)
            self.width_all = self._adapt(self.width_all, accept)
            self.width_real = self.width_all.copy()
            self.width_complex = self.width_all.copy()
        elif (group == "all" and self.mode == "real") or group == "real":
            self.width_real = self._adapt(self.width_real, accept)          # :440-446
        else:
            self.width_complex = self._adapt(self.width_complex, accept)    # :449-456
        self.step_index += 1

    def step_magnitude_phase(self, n_sweeps=1):
        """``step_complex_group`` under ``complex_sample_method="magnitude-phase"`` (:168-207, :304-317).

        Stream words of one step: Box-Muller pairs ``0 .. W1-1`` (``W1 = 2 ceil(nc/2)``) give the ``nc`` magnitude
        normals, word ``W1`` the accept draw of the magnitude stage, words ``W1+1 .. W1+nc`` the phases
        (``-pi + 2 pi u``) and word ``W1+nc+1`` the accept draw of the phase stage.
        """
        nr, nc = self.nr, self.nc
        for _ in range(n_sweeps):
            w1 = philox.n_normal_words(nc)
            words = philox.step_words(self.seed, self.chain_ids, self.step_index, w1 + nc + 2)
            unit = philox.unit_open(words)
            r = np.sqrt(-2.0 * np.log(unit[:, 0:w1:2]))
            theta = (2.0 * np.pi) * unit[:, 1:w1:2]
            g = np.empty((self.n_chains, w1))
            g[:, 0::2] = r * np.cos(theta)
            g[:, 1::2] = r * np.sin(theta)
            g = g[:, :nc]
            # --- magnitude stage: m' = m + (sigma_c^2 Re K_jj) g at fixed phase; the width adapts (:178-192)
            z = self.complex_params()
            mag = np.abs(z)
            direction = np.where(mag > 0, z / np.where(mag > 0, mag, 1.0), 1.0 + 0j)       # polar(0) = (0, phase 0)
            spread = self.width_complex[:, None] ** 2 * np.real(np.einsum("njj->nj", self.cov_complex))
            znew = (mag + spread * g) * direction
            prop = self.x.copy()
            prop[:, nr:nr + nc], prop[:, nr + nc:] = znew.real, znew.imag
            accept = self._decide(prop, unit[:, w1])
            self.width_complex = self._adapt(self.width_complex, accept)
            # --- phase stage: every phase redrawn in (-pi, pi) at fixed magnitude; no width update (:194-207)
            z = self.complex_params()
            phases = -np.pi + 2.0 * np.pi * unit[:, w1 + 1:w1 + 1 + nc]
            znew = np.abs(z) * (np.cos(phases) + 1j * np.sin(phases))
            prop = self.x.copy()
            prop[:, nr:nr + nc], prop[:, nr + nc:] = znew.real, znew.imag
            self._decide(prop, unit[:, w1 + nc + 1])
            self.step_index += 1

    # ------------------------------------------------------------------ measure
    def measure(self):
        nr, nc = self.nr, self.nc
        self.measure_step_counter += 1
        n = self.measure_step_counter
        old = self.mean.copy()
        self.mean = self.mean * ((n - 1) / n) + self.x / n
        if n > MEASURES_BEFORE_COVARIANCE:
            if nr:
                o, mu, xr = old[:, :nr], self.mean[:, :nr], self.x[:, :nr]
                eps = self.width_real ** 2 / n
                self.cov_real = (self.cov_real * ((n - 2) / (n - 1))
                                 + (np.einsum("ni,nj->nij", o, o) - n / (n - 1) * np.einsum("ni,nj->nij", mu, mu)
                                    + np.einsum("ni,nj->nij", xr, xr) / (n - 1)
                                    + eps[:, None, None] * np.identity(nr)))
            if nc:
                def cplx(a):
                    return a[:, nr:nr + nc] + 1j * a[:, nr + nc:]
                o, mu, z = cplx(old), cplx(self.mean), cplx(self.x)
                eps = self.width_complex ** 2 / n
                self.cov_complex = (self.cov_complex * ((n - 2) / (n - 1))
                                    + (np.einsum("ni,nj->nij", o, o.conj())
                                       - n / (n - 1) * np.einsum("ni,nj->nij", mu, mu.conj())
                                       + np.einsum("ni,nj->nij", z, z.conj()) / (n - 1)
                                       + eps[:, None, None] * np.identity(nc)))
            if self.adapt_shape:
                self._refresh_factors()
        self.observables_mean = self.observables_mean * ((n - 1) / n) + self.observables() / n

    # ------------------------------------------------------------------ pooled moments
    def pooled_moments(self):
        """``[n, sum x (D), sum x x^T packed lower (D(D+1)/2), sum obs (2nr+nc), accepted, proposed]``."""
        d = self.dim
        il = np.tril_indices(d)
        xx = np.einsum("ni,nj->ij", self.x, self.x)
        return np.concatenate(([float(self.n_chains)], self.x.sum(axis=0), xx[il], self.observables().sum(axis=0),
                               [float(self.accepted), float(self.proposed)]))
