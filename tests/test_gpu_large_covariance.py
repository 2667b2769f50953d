"""GPU tests of the statistics-only per-chain covariance of large parameter spaces (64 real parameters: 2 080 packed
entries per chain, ME_FLAG_TRACK_COVARIANCE): update_covariance_matrix_real (metropolis_engine.py:416-421) streamed
through k_measure, checked against the many-chain oracle, and addressed correctly beyond 4 GiB per field."""
import numpy as np
import pytest

import metropolisengine_amd as me
from metropolisengine_amd import _capi
from metropolisengine_amd.engine import unpack_real_block
from oracle import energies
from oracle.manychain import ManyChainOracle

pytestmark = pytest.mark.gpu

_M = np.random.default_rng(5).standard_normal((64, 64))
AMAT = _M @ _M.T / 64 + np.identity(64)


def test_default_keeps_no_large_covariance():
    eng = me.MetropolisEngine(me.DenseQuadratic(AMAT), None, [0.0] * 64, None, temp=1.0, n_chains=128, cov_mode="fixed")
    eng.measure()
    with pytest.raises(NotImplementedError):
        eng.covariance_matrix_real


def test_streamed_covariance_matches_oracle_f64():
    n, seed = 64 + 7, 3
    x0 = list(np.linspace(-0.2, 0.2, 64))
    eng = me.MetropolisEngine(me.DenseQuadratic(AMAT), None, x0, None, temp=1.0, n_chains=n, seed=seed, dtype="f64",
                              sampling_width=0.1, cov_mode="fixed", track_covariance=True)
    ora = ManyChainOracle(64, 0, energies.dense_quadratic(64, 0, AMAT), n, seed=seed, temp=1.0,
                          initial_real_params=x0, sampling_width=0.1, adapt_shape=False)
    for _ in range(56):                 # the covariance starts moving after 50 measures (:389)
        eng.step_all(2)
        ora.step(2)
        eng.measure()
        ora.measure()
    assert np.allclose(eng._get(_capi.FIELD_PARAMS), ora.x, rtol=0, atol=1e-9)
    cov = eng.covariance_matrix_real
    assert cov.shape == (n, 64, 64)
    assert np.allclose(cov, ora.cov_real, rtol=0, atol=1e-9)
    assert not np.allclose(cov[0], np.identity(64), atol=1e-3)      # it has been updated
    assert np.allclose(eng.real_mean, ora.mean[:, :64], rtol=0, atol=1e-9)


def test_rows_beyond_4_gib_are_addressed_correctly():
    """2^19 chains x 2 080 floats = 4.36 GB in one field: the last chains of the big engine equal a small engine that
    owns the same global chain ids (same Philox streams, same arithmetic -> bitwise in float32)."""
    n, tail = 1 << 19, 96
    kw = dict(temp=1.0, seed=9, sampling_width=0.1, cov_mode="fixed", track_covariance=True)
    big = me.MetropolisEngine(me.DenseQuadratic(AMAT), None, [0.0] * 64, None, n_chains=n, **kw)
    small = me.MetropolisEngine(me.DenseQuadratic(AMAT), None, [0.0] * 64, None, n_chains=tail,
                                chain_offset=n - tail, **kw)
    for eng in (big, small):
        for _ in range(53):
            eng.step_all(1)
            eng.measure()
    a = big._get(_capi.FIELD_COV, n - tail, tail)
    b = small._get(_capi.FIELD_COV)
    assert a.shape == (tail, 2080)
    assert np.array_equal(a, b)
    first = big._get(_capi.FIELD_COV, 0, 4)
    assert np.all(np.isfinite(first)) and not np.array_equal(first[0], a[0])
    diag = unpack_real_block(a, 64)[:, np.arange(64), np.arange(64)]
    assert np.all(diag > 0)


def test_streamed_covariance_with_a_complex_block_f64():
    """1 real + 13 complex parameters: 170 packed entries, just past the factor kernels; the kernel set is compiled on
    demand (build_dims) and the Hermitian block goes through the same streaming path."""
    nr, nc, n, seed = 1, 13, 37, 21
    a, b = (0.7,), tuple(0.5 + 0.25 * j for j in range(nc))
    real0, cplx0 = [0.1], [0.05 * (j + 1) * np.exp(0.3j * j) for j in range(nc)]
    eng = me.MetropolisEngine(me.DiagQuadratic(a, b), None, real0, cplx0, temp=1.0, n_chains=n, seed=seed, dtype="f64",
                              sampling_width=0.2, cov_mode="fixed", track_covariance=True)
    ora = ManyChainOracle(nr, nc, energies.diag_quadratic(nr, nc, a, b), n, seed=seed, temp=1.0,
                          initial_real_params=real0, initial_complex_params=cplx0, sampling_width=0.2,
                          adapt_shape=False)
    for _ in range(57):
        eng.step_all(3)
        ora.step(3)
        eng.measure()
        ora.measure()
    assert np.allclose(eng._get(_capi.FIELD_PARAMS), ora.x, rtol=0, atol=1e-9)
    assert np.allclose(eng.covariance_matrix_real, ora.cov_real, rtol=0, atol=1e-9)
    assert np.allclose(eng.covariance_matrix_complex, ora.cov_complex, rtol=0, atol=1e-9)
    assert np.abs(ora.cov_complex[0] - np.identity(nc)).max() > 1e-3
    assert eng.accept_stats() == (ora.accepted, ora.proposed)
