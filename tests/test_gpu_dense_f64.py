"""GPU tests of the float64 matrix-core path for BASELINE config 4 (64 real parameters, E = x^T A x at the
reference's precision): k_step_dense64_f64 (v_mfma_f64_16x16x4_f64 on the folded lower triangle, me_dense_f64.h)
follows the float64 oracle as a TRAJECTORY -- identity shape and the shared-factor proposal x' = x + sigma L g
(metropolis_engine.py:225-239, :261-272)."""
import numpy as np
import pytest

import metropolisengine_amd as me
from metropolisengine_amd.distributed import moments_to_statistics
from oracle import energies
from oracle.manychain import ManyChainOracle

pytestmark = pytest.mark.gpu

_M = np.random.default_rng(5).standard_normal((64, 64))
AMAT = _M @ _M.T / 64 + np.identity(64)
ASYM = AMAT + 0.05 * np.triu(np.random.default_rng(6).standard_normal((64, 64)), 1)   # only its symmetric part may matter


def _packed_lower(mat):
    return mat[np.tril_indices(mat.shape[0])]


@pytest.mark.parametrize("matrix", [AMAT, ASYM], ids=["spd", "asymmetric"])
@pytest.mark.parametrize("fused", [1, 5], ids=["one-sweep-launches", "fused-sweeps"])
def test_dense64_f64_identity_shape_follows_oracle(matrix, fused):
    n, seed, sweeps = 64 * 3 + 5, 21, 60            # 197 chains: ragged last wavefront (shadow lanes)
    x0 = list(np.linspace(-0.3, 0.3, 64))
    eng = me.MetropolisEngine(me.DenseQuadratic(matrix), None, x0, None, temp=1.0, n_chains=n, seed=seed,
                              sampling_width=0.1, cov_mode="fixed", dtype="f64")
    ora = ManyChainOracle(64, 0, energies.dense_quadratic(64, 0, matrix), n, seed=seed, temp=1.0,
                          initial_real_params=x0, sampling_width=0.1, adapt_shape=False)
    assert np.allclose(eng.energy_total, ora.energy, rtol=1e-13)
    for _ in range(sweeps // fused):
        eng.step_all(fused)
        ora.step(fused)
    assert np.allclose(eng._get(0), ora.x, rtol=0, atol=1e-9)
    assert np.allclose(eng.energy_total, ora.energy, rtol=0, atol=1e-9)
    assert np.allclose(eng.real_group_sampling_width, ora.width_real, rtol=1e-12)
    assert eng.accept_stats() == (ora.accepted, ora.proposed)


def test_dense64_f64_shared_factor_follows_oracle():
    n, seed, sweeps = 64 * 4 + 17, 22, 50
    rng = np.random.default_rng(9)
    b = rng.standard_normal((64, 64))
    cov = 0.5 * np.linalg.inv(AMAT) + 0.02 * (b @ b.T) / 64           # some SPD proposal covariance
    x0 = list(np.linspace(0.2, -0.2, 64))
    eng = me.MetropolisEngine(me.DenseQuadratic(AMAT), None, x0, None, temp=1.0, n_chains=n, seed=seed,
                              sampling_width=0.3, cov_mode="pooled", dtype="f64")
    eng.set_shared_factor(_packed_lower(np.linalg.cholesky(cov)))
    ora = ManyChainOracle(64, 0, energies.dense_quadratic(64, 0, AMAT), n, seed=seed, temp=1.0,
                          initial_real_params=x0, sampling_width=0.3, covariance_matrix_real=cov, adapt_shape=False)
    for _ in range(sweeps // 2):
        eng.step_all(2)
        ora.step(2)
    assert np.allclose(eng._get(0), ora.x, rtol=0, atol=1e-9)
    assert np.allclose(eng.energy_total, ora.energy, rtol=0, atol=1e-9)
    assert np.allclose(eng.real_group_sampling_width, ora.width_real, rtol=1e-12)
    assert eng.accept_stats() == (ora.accepted, ora.proposed)
    assert 0.05 < ora.accepted / ora.proposed < 0.9


def test_dense64_f64_hard_wall_and_zero_temperature():
    n, seed = 130, 23
    x0 = [0.9] + [0.0] * 63
    for temp, wall in ((0.0, None), (1.0, me.AbsReal0AtLeast(1.0))):
        eng = me.MetropolisEngine(me.DenseQuadratic(AMAT), wall, x0, None, temp=temp, n_chains=n, seed=seed,
                                  sampling_width=0.2, cov_mode="fixed", dtype="f64")
        ora = ManyChainOracle(64, 0, energies.dense_quadratic(64, 0, AMAT), n, seed=seed, temp=temp,
                              initial_real_params=x0, sampling_width=0.2, adapt_shape=False,
                              reject=None if wall is None else (lambda p: np.abs(p[:, 0]) >= 1.0))
        eng.step_all(25)
        ora.step(25)
        assert np.allclose(eng._get(0), ora.x, rtol=0, atol=1e-9)
        assert eng.accept_stats() == (ora.accepted, ora.proposed)
        assert np.all(np.abs(eng._get(0)[:, 0]) < 1.0)


def test_dense64_f64_full_size_shard_equivalence_and_moments():
    """2^19 chains (BASELINE config 4): a sub-range of the full-size engine is bitwise equal to a small engine owning
    the same global chain ids, and the ensemble covariance is T/2 A^-1 within the ensemble error."""
    n = 1 << 19
    kw = dict(temp=1.0, seed=2026, sampling_width=0.2, cov_mode="fixed", dtype="f64")
    big = me.MetropolisEngine(me.DenseQuadratic(AMAT), None, [0.0] * 64, None, n_chains=n, **kw)
    lo, cnt = 64 * 4001 + 7, 300
    small = me.MetropolisEngine(me.DenseQuadratic(AMAT), None, [0.0] * 64, None, n_chains=cnt, chain_offset=lo, **kw)
    big.step_all(40)
    small.step_all(40)
    assert np.array_equal(big._get(0)[lo:lo + cnt], small._get(0))
    assert np.array_equal(big.energy_total[lo:lo + cnt], small.energy_total)
    big.step_all(1500)
    st = moments_to_statistics(big.pooled_moments(), 64, 0)
    want = 0.5 * np.linalg.inv(AMAT)
    assert np.all(np.abs(st["covariance"] - want) < 8 * np.max(np.abs(want)) * np.sqrt(2.0 / n) * 6)
    assert 0.15 < st["acceptance_rate"] < 0.5


@pytest.mark.parametrize("n", [1, 31, 33, 95])
def test_dense64_f64_ragged_tiles_are_shards_of_a_larger_engine(n):
    """Chain counts that leave a wavefront's 32-chain tile (two chains per lane) partly or almost wholly empty: the n chains
    are bit for bit the first n of a 128-chain engine, with the identity shape and with a shared factor."""
    kw = dict(temp=1.0, seed=77, sampling_width=0.15, dtype="f64")
    x0 = list(np.linspace(-0.2, 0.4, 64))
    factor = _packed_lower(np.linalg.cholesky(0.5 * np.linalg.inv(AMAT)))
    for mode in ("fixed", "pooled"):
        big = me.MetropolisEngine(me.DenseQuadratic(AMAT), None, x0, None, n_chains=128, cov_mode=mode, **kw)
        small = me.MetropolisEngine(me.DenseQuadratic(AMAT), None, x0, None, n_chains=n, cov_mode=mode, **kw)
        if mode == "pooled":
            big.set_shared_factor(factor)
            small.set_shared_factor(factor)
        for fused in (1, 3, 1):
            big.step_all(fused)
            small.step_all(fused)
        assert np.array_equal(big._get(0)[:n], small._get(0))
        # (a one-chain engine reports scalars, as the reference does)
        assert np.array_equal(big.energy_total[:n], np.atleast_1d(small.energy_total))
        assert np.array_equal(big.real_group_sampling_width[:n], np.atleast_1d(small.real_group_sampling_width))
        assert small.accept_stats()[1] == 5 * n


def test_pooled_moments_float64_gram_kernel_matches_numpy():
    """64 real parameters in float64 pool their second moments with v_mfma_f64_16x16x4_f64 (me_pool_gram.h,
    k_pool_gram64_f64): the result is the plain float64 sums over the chains' current states -- ragged last tile, several
    tiles per wavefront, fewer tiles than wavefronts -- to rounding, and bitwise reproducible."""
    from metropolisengine_amd.distributed import moments_size
    for n in (5, 64 * 3 + 5, (1 << 16) + 64 * 7 + 9):
        eng = me.MetropolisEngine(me.DenseQuadratic(AMAT), None, list(np.linspace(-0.5, 0.5, 64)), None, temp=1.0,
                                  n_chains=n, seed=13, sampling_width=0.2, cov_mode="fixed", dtype="f64")
        eng.step_all(40)
        x = eng._get(0)
        m = eng.pooled_moments()
        assert m.shape == (moments_size(64, 0),)
        acc, prop = eng.accept_stats()
        assert m[0] == n and m[-2] == acc and m[-1] == prop
        assert np.allclose(m[1:65], x.sum(axis=0), rtol=1e-12, atol=1e-11)
        second = x.T @ x
        assert np.allclose(m[65:65 + 2080], second[np.tril_indices(64)], rtol=1e-12, atol=1e-11)
        obs = m[65 + 2080:65 + 2080 + 128]
        assert np.allclose(obs[:64], np.abs(x).sum(axis=0), rtol=1e-12)
        assert np.allclose(obs[64:], (x * x).sum(axis=0), rtol=1e-12)
        assert np.array_equal(m, eng.pooled_moments())             # fixed summation order: bitwise reproducible
