"""A parameter space beyond the register-resident kernels: 150 real parameters coupled in a chain (a discretised
elastic string, E = sum_i a x_i^2 + k (x_{i+1} - x_i)^2), sampled with the reference's default semantics -- every chain adapts
its own 150 x 150 proposal covariance after 50 measures (metropolis_engine.py:416-421 feeding :261-272) -- and then with
one shape pooled over the whole ensemble, which is what thousands of chains on a GPU add to the reference's scheme.

The reference has no limit on the number of parameters (:41-60); here this size runs on the runtime-dimension kernel set
(csrc/me_runtime_dims.hip: no kernel build, the dimensions are launch arguments).  The ensemble covariance is compared with
the exact one, T/2 A^-1.

    python examples/demo_large_space.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metropolisengine_amd as me                                   # noqa: E402
from metropolisengine_amd.distributed import adapt_pooled_shape, moments_to_statistics  # noqa: E402


def string_matrix(d, a=1.0, k=0.5):
    """A with x^T A x = sum_i a x_i^2 + k sum_i (x_{i+1} - x_i)^2 (tridiagonal, positive definite)."""
    m = np.zeros((d, d))
    for i in range(d):
        m[i, i] = a + k * ((i > 0) + (i < d - 1))
        if i + 1 < d:
            m[i, i + 1] = m[i + 1, i] = -k
    return m


def deviation(eng, exact, d):
    stats = moments_to_statistics(eng.pooled_moments(), d, 0)
    return np.max(np.abs(stats["covariance"] - exact)) / np.max(np.abs(exact)), stats


def main(d=150, n_chains=1 << 12, cycles=90, sweeps_per_cycle=20, temp=1.0, dtype="f64", coupling=0.5, width=0.08):
    amat = string_matrix(d, k=coupling)
    exact = 0.5 * temp * np.linalg.inv(amat)
    kw = dict(initial_real_params=[0.0] * d, temp=temp, n_chains=n_chains, seed=7, sampling_width=width, dtype=dtype)

    # 1. the reference's semantics (cov_mode="reference", the default): per-chain running covariances shape the proposals
    per_chain = me.MetropolisEngine(me.DenseQuadratic(amat), **kw)
    for c in range(cycles):
        per_chain.cycle(sweeps_per_cycle)                                # = sweeps_per_cycle x step_all(); measure()
    err_ref, _ = deviation(per_chain, exact, d)
    print("%d parameters x %d chains, %d sweeps, each chain adapting its own %d x %d shape: acceptance %.2f, ensemble covariance "
          "within %.0f %% (of its largest entry) of T/2 A^-1" % (d, n_chains, cycles * sweeps_per_cycle, d, d, per_chain.acceptance_rate(), 100 * err_ref))
    print("  (a chain's own history of a few dozen correlated samples spans few of the %d directions: the reference's recursion "
          "is reproduced as it is -- oracle parity -- not improved upon)" % d)

    # 2. what many chains make possible: ONE shape estimated from the whole ensemble (cov_mode="pooled")
    pooled = me.MetropolisEngine(me.DenseQuadratic(amat), cov_mode="pooled", **kw)
    for c in range(cycles):
        pooled.step_all(sweeps_per_cycle)
        if c % 10 == 9:
            adapt_pooled_shape(pooled)                                   # ensemble moments -> Cholesky -> shared factor
    err_pool, stats = deviation(pooled, exact, d)
    print("the same budget with one shape pooled over the ensemble every %d sweeps: acceptance %.2f, ensemble covariance within "
          "%.0f %%" % (10 * sweeps_per_cycle, pooled.acceptance_rate(), 100 * err_pool))
    return per_chain, pooled, err_ref, err_pool


if __name__ == "__main__":
    main()
