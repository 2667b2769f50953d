"""GPU tests of the matrix-core path for BASELINE config 4 (64 real parameters, E = x^T A x, float32):
k_step_dense64_bf16x3 (default; split-bf16 on the matrix pipe) and k_step_dense64_mfma (METROPOLIS_DENSE64_FP32_MFMA=1)
against the float64 oracle (one step) and against the analytic stationary covariance T/2 A^-1."""
import os
import subprocess
import sys

import numpy as np
import pytest

import metropolisengine_amd as me
from metropolisengine_amd.distributed import adapt_pooled_shape, moments_to_statistics
from oracle import energies
from oracle.manychain import ManyChainOracle

pytestmark = pytest.mark.gpu

_M = np.random.default_rng(5).standard_normal((64, 64))
AMAT = _M @ _M.T / 64 + np.identity(64)
ASYM = AMAT + 0.05 * np.triu(np.random.default_rng(6).standard_normal((64, 64)), 1)   # not symmetric: catches a transposed operand


# a matrix with entries spread over six decades exercises the three-piece split of both operands
WIDE = AMAT * np.exp(np.random.default_rng(7).uniform(-7.0, 7.0, size=(64, 64)))
WIDE = 0.5 * (WIDE + WIDE.T) + 40.0 * np.diag(np.abs(WIDE).sum(axis=1)) / 64


@pytest.mark.parametrize("matrix,temp", [(AMAT, 1.0), (ASYM, 1.0), (AMAT * 3e4, 3e4), (WIDE, 50.0)],
                         ids=["spd", "asymmetric", "scaled", "wide-range"])
def test_mfma_energy_one_step_vs_oracle(matrix, temp):
    n, seed = 4096 + 37, 11          # ragged tail: the last wavefront has shadow lanes
    x0 = list(np.linspace(-0.3, 0.3, 64))
    eng = me.MetropolisEngine(me.DenseQuadratic(matrix), None, x0, None, temp=temp, n_chains=n, seed=seed,
                              sampling_width=0.1, cov_mode="fixed")
    ora = ManyChainOracle(64, 0, energies.dense_quadratic(64, 0, matrix), n, seed=seed, temp=temp,
                          initial_real_params=x0, sampling_width=0.1)
    assert np.allclose(eng.energy_total, ora.energy, rtol=2e-6)
    eng.step_all()
    ora.step()
    x, e = eng._get(0), eng.energy_total
    same = np.all(np.abs(x - ora.x) < 2e-5, axis=1)
    assert same.mean() > 0.998                              # float32 accept flips only on near-ties
    assert np.allclose(e[same], ora.energy[same], rtol=1e-5)
    acc, prop = eng.accept_stats()
    assert prop == n and abs(acc - ora.accepted) <= 8
    eng.step_all(3)                                         # fused sweeps run the same code
    ora.step(3)
    same = np.all(np.abs(eng._get(0) - ora.x) < 1e-4, axis=1)
    assert same.mean() > 0.99


@pytest.mark.parametrize("shape", ["identity", "shared"])
def test_split_bf16_kernel_50_sweep_trajectory_against_the_oracle(shape):
    """The float32 matrix-core kernel tied DIRECTLY to the float64 oracle over a trajectory (round 2 tied it to the float64
    kernel only): same Philox streams, 50 sweeps, a chain's accept sequence must equal the oracle's for >= 98 % of the
    chains (the rest are near-ties of u against exp(-dE/T) that a 1e-7 relative error in dE flips), and those chains' states,
    energies and adapted widths agree to float32 accuracy.  Identity shape and a shared proposal factor (L g on the cores)."""
    n, seed, sweeps = 2048 + 19, 23, 50
    x0 = list(np.linspace(-0.2, 0.2, 64))
    kw = dict(temp=1.0, n_chains=n, seed=seed, sampling_width=0.08)
    okw = dict(seed=seed, temp=1.0, initial_real_params=x0, sampling_width=0.08, adapt_shape=False)
    if shape == "shared":
        b = np.random.default_rng(9).standard_normal((64, 64))
        cov = 0.5 * np.linalg.inv(AMAT) + 0.05 * (b @ b.T) / 64
        eng = me.MetropolisEngine(me.DenseQuadratic(AMAT), None, x0, None, cov_mode="pooled", **kw)
        eng.set_shared_factor(np.linalg.cholesky(cov)[np.tril_indices(64)])
        ora = ManyChainOracle(64, 0, energies.dense_quadratic(64, 0, AMAT), n, covariance_matrix_real=cov, **okw)
    else:
        eng = me.MetropolisEngine(me.DenseQuadratic(AMAT), None, x0, None, cov_mode="fixed", **kw)
        ora = ManyChainOracle(64, 0, energies.dense_quadratic(64, 0, AMAT), n, **okw)
    same = np.ones(n, dtype=bool)
    for _ in range(sweeps):
        before = ora.accepted
        eng.step_all()
        ora.step()
        moved_gpu = np.any(np.abs(eng._get(0) - ora.x) > 1e-3, axis=1)        # a flipped decision moves the state by ~width
        same &= ~moved_gpu
        assert ora.accepted >= before
    assert same.mean() >= 0.98, same.mean()
    x = eng._get(0)
    assert np.max(np.abs(x[same] - ora.x[same])) < 2e-4
    assert np.allclose(eng.energy_total[same], ora.energy[same], rtol=2e-5, atol=2e-5)
    assert np.allclose(eng.sampling_width[same], ora.width_real[same], rtol=2e-5)
    acc, prop = eng.accept_stats()
    assert prop == n * sweeps and abs(acc - ora.accepted) <= 0.02 * n + 8
    assert 0.05 < ora.accepted / ora.proposed < 0.95


def test_mfma_stationary_covariance_identity_and_pooled():
    n = 1 << 13
    want = 0.5 * np.linalg.inv(AMAT)                        # density exp(-x^T A x / T), T = 1
    eng = me.MetropolisEngine(me.DenseQuadratic(AMAT), None, [0.0] * 64, None, temp=1.0, n_chains=n, seed=4,
                              sampling_width=0.2, cov_mode="pooled")
    eng.step_all(3000)                                      # identity proposal shape so far
    st = moments_to_statistics(eng.pooled_moments(), 64, 0)
    tol = 6 * np.max(np.abs(want)) * np.sqrt(2.0 / n)
    assert np.all(np.abs(st["covariance"] - want) < tol)
    rate_identity = st["acceptance_rate"]
    # pooled_shared: one Cholesky factor of the pooled covariance shapes every chain's proposals (L g on the MFMA)
    pooled = adapt_pooled_shape(eng)                        # pooled_statistics -> pooled_factor -> set_shared_factor
    assert np.allclose(pooled["covariance"], st["covariance"])
    eng.step_all(2000)
    st2 = moments_to_statistics(eng.pooled_moments(), 64, 0)
    assert np.all(np.abs(st2["covariance"] - want) < tol)
    assert 0.15 < rate_identity < 0.5 and 0.15 < st2["acceptance_rate"] < 0.5
    assert np.all(np.abs(st2["mean"]) < 6 * np.sqrt(np.max(np.diag(want)) / n))


def test_mfma_matches_f64_kernel_statistics():
    """The float64 (VALU) kernels run the same configuration; pooled second moments agree within MC error."""
    n = 1 << 11
    kw = dict(temp=1.0, n_chains=n, seed=8, sampling_width=0.2, cov_mode="fixed")
    f32 = me.MetropolisEngine(me.DenseQuadratic(AMAT), None, [0.0] * 64, None, **kw)
    f64 = me.MetropolisEngine(me.DenseQuadratic(AMAT), None, [0.0] * 64, None, dtype="f64", **kw)
    f32.step_all(1500)
    f64.step_all(1500)
    a = moments_to_statistics(f32.pooled_moments(), 64, 0)
    b = moments_to_statistics(f64.pooled_moments(), 64, 0)
    assert abs(a["acceptance_rate"] - b["acceptance_rate"]) < 0.01
    assert np.all(np.abs(np.diag(a["covariance"]) - np.diag(b["covariance"])) < 8 * 0.5 * np.sqrt(2.0 / n))


def test_fp32_mfma_variant_in_child_process():
    """The kernel choice is read once per process: run the one-step and f64-statistics checks again under
    METROPOLIS_DENSE64_FP32_MFMA=1 (one child test process)."""
    if os.environ.get("METROPOLIS_DENSE64_FP32_MFMA") == "1":
        pytest.skip("already the fp32 MFMA variant")
    env = dict(os.environ, METROPOLIS_DENSE64_FP32_MFMA="1")
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                          "one_step or f64_kernel or gram", "-p", "no:cacheprovider"], env=env, capture_output=True, text=True,
                         timeout=600, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    assert "6 passed" in res.stdout


def test_pooled_moments_gram_kernel_matches_numpy():
    """64 real parameters pool their second moments with the matrix-core Gram kernel (me_pool_gram.h); the result must be
    the plain sums over the chains' current states (ragged last tile, several tiles per wavefront)."""
    from metropolisengine_amd.distributed import moments_size
    for n in (64 * 3 + 5, (1 << 16) + 64 * 7 + 9):
        eng = me.MetropolisEngine(me.DenseQuadratic(AMAT), None, list(np.linspace(-0.5, 0.5, 64)), None, temp=1.0,
                                  n_chains=n, seed=13, sampling_width=0.2, cov_mode="fixed")
        eng.step_all(40)
        x = eng._get(0)
        m = eng.pooled_moments()
        assert m.shape == (moments_size(64, 0),)
        acc, prop = eng.accept_stats()
        assert m[0] == n and m[-2] == acc and m[-1] == prop
        assert np.allclose(m[1:65], x.sum(axis=0), rtol=5e-6, atol=1e-3)
        second = x.T @ x
        assert np.allclose(m[65:65 + 2080], second[np.tril_indices(64)], rtol=5e-6, atol=1e-3 * np.sqrt(n))
        obs = m[65 + 2080:65 + 2080 + 128]
        assert np.allclose(obs[:64], np.abs(x).sum(axis=0), rtol=5e-6)
        assert np.allclose(obs[64:], (x * x).sum(axis=0), rtol=5e-6)
        again = eng.pooled_moments()
        assert np.array_equal(m, again)             # fixed summation order: bitwise reproducible
