"""GPU tests of the runtime-dimension kernel set (csrc/me_runtime_dims.hip): parameter spaces beyond the 96 real degrees
of freedom of the register-resident kernels.  The reference has no limit on the number of parameters
(metropolis_engine.py:41-60).  float64 follows the many-chain oracle on the same Philox streams (1e-9): step_all, fused
sweeps, group-wise steps of a mixed engine, the hard wall, T = 0, measure(); float32 is checked on stationary moments."""
import numpy as np
import pytest

import metropolisengine_amd as me
from metropolisengine_amd import build
from oracle import energies
from oracle.manychain import ManyChainOracle

pytestmark = pytest.mark.gpu
TOL = 1e-9


def _compare(eng, ora, nr, nc):
    assert np.allclose(eng._get(0), ora.x, rtol=0, atol=TOL)
    assert np.allclose(eng.energy_total, ora.energy, rtol=0, atol=1e-8)
    if nr:
        assert np.allclose(eng.real_group_sampling_width, ora.width_real, rtol=1e-12)
    if nc:
        assert np.allclose(eng.complex_group_sampling_width, ora.width_complex, rtol=1e-12)
    assert np.allclose(eng._get(3), ora.mean, rtol=0, atol=TOL)
    assert np.allclose(eng.observables_mean, ora.observables_mean, rtol=0, atol=TOL)
    assert eng.accept_stats() == (ora.accepted, ora.proposed)


@pytest.mark.parametrize("temp", [1.0, 0.0], ids=["T1", "T0"])
def test_130_real_isotropic_follows_the_oracle(temp):
    nr, n, seed = 130, 197, 31                      # 197 chains: ragged last wavefront
    x0 = list(np.linspace(-0.2, 0.2, nr))
    eng = me.MetropolisEngine(me.IsoQuadratic(0.7), None, x0, None, temp=temp, n_chains=n, seed=seed, dtype="f64",
                              cov_mode="fixed", sampling_width=0.03)
    ora = ManyChainOracle(nr, 0, energies.iso_quadratic(nr, 0, 0.7), n, seed=seed, temp=temp, initial_real_params=x0,
                          sampling_width=0.03, adapt_shape=False)
    for k in range(20):
        sweeps = 1 if k % 2 else 3                  # one-sweep launches and fused sweeps
        eng.step_all(sweeps)
        ora.step(sweeps)
        eng.measure()
        ora.measure()
    _compare(eng, ora, nr, 0)
    assert 0.05 < ora.accepted / ora.proposed < 0.95 or temp == 0.0


def test_mixed_60_real_25_complex_groups_and_wall():
    nr, nc, n, seed = 60, 25, 130, 32               # D = 110; an odd number of real parameters would shift the pairs
    rng = np.random.default_rng(3)
    a, b = rng.uniform(0.5, 3.0, nr), rng.uniform(0.5, 3.0, nc)
    x0 = [0.9] + [0.0] * (nr - 1)
    z0 = list(0.05 * (rng.standard_normal(nc) + 1j * rng.standard_normal(nc)))
    eng = me.MetropolisEngine(me.DiagQuadratic(tuple(a), tuple(b)), me.AbsReal0AtLeast(1.0), x0, z0, temp=0.5, n_chains=n,
                              seed=seed, dtype="f64", cov_mode="fixed", sampling_width=0.04)
    ora = ManyChainOracle(nr, nc, energies.diag_quadratic(nr, nc, a, b), n, seed=seed, temp=0.5, initial_real_params=x0,
                          initial_complex_params=z0, sampling_width=0.04, adapt_shape=False,
                          reject=lambda p: np.abs(p[:, 0]) >= 1.0)
    for k in range(12):
        for op in ("all", "real", "complex", "all"):
            if op == "all":
                eng.step_all(2)
                ora.step(2, group="all")
            elif op == "real":
                eng.step_real_group()
                ora.step(1, group="real")
            else:
                eng.step_complex_group()
                ora.step(1, group="complex")
        eng.measure()
        ora.measure()
    _compare(eng, ora, nr, nc)
    assert np.allclose(eng.sampling_width, ora.width_all, rtol=1e-12)
    assert np.all(np.abs(eng._get(0)[:, 0]) < 1.0)


def test_odd_dimension_101_real():
    """D odd: the last Box-Muller pair is half used and the accept uniform is word D + 1."""
    nr, n, seed = 101, 70, 33
    x0 = [0.1] * nr
    eng = me.MetropolisEngine(me.IsoQuadratic(1.0), None, x0, None, temp=1.0, n_chains=n, seed=seed, dtype="f64",
                              cov_mode="fixed")
    ora = ManyChainOracle(nr, 0, energies.iso_quadratic(nr, 0, 1.0), n, seed=seed, temp=1.0, initial_real_params=x0, adapt_shape=False)
    eng.step_all(30)
    ora.step(30)
    assert np.allclose(eng._get(0), ora.x, rtol=0, atol=TOL)
    assert eng.accept_stats() == (ora.accepted, ora.proposed)


def test_f32_200_real_stationary_variances():
    nr, n = 200, 4096
    a = np.random.default_rng(4).uniform(0.5, 4.0, nr)
    eng = me.MetropolisEngine(me.DiagQuadratic(tuple(a)), None, [0.0] * nr, None, temp=1.0, n_chains=n, seed=34,
                              cov_mode="fixed", sampling_width=0.05)
    eng.step_all(4000)
    x = eng._get(0)
    var = x.var(axis=0)
    assert np.all(np.abs(var * 2 * a - 1) < 8 * np.sqrt(2.0 / n))          # Var x_i = T / (2 a_i)
    assert 0.15 < eng.acceptance_rate() < 0.5
    assert np.allclose(eng.energy_total, (a * x * x).sum(axis=1), rtol=2e-4)   # the ledger is the energy of the state


@pytest.mark.parametrize("shape", ["identity", "shared"])
def test_dense_energy_and_shared_factor_at_120_real(shape):
    """Beyond 96 degrees of freedom a dense quadratic form (x' parked in LDS, folded triangle through wave-uniform loads)
    and one shared proposal factor (cov_mode="pooled", pure real spaces) follow the oracle: an asymmetric A (only its
    symmetric part may matter), fused and one-sweep launches, the wall."""
    nr, n, seed = 120, 100, 35
    rng = np.random.default_rng(8)
    m = rng.standard_normal((nr, nr))
    amat = m @ m.T / nr + np.identity(nr) + 0.05 * np.triu(rng.standard_normal((nr, nr)), 1)
    x0 = [0.5] + list(np.linspace(-0.1, 0.1, nr - 1))
    kw = dict(temp=1.0, n_chains=n, seed=seed, dtype="f64", sampling_width=0.05)
    okw = dict(seed=seed, temp=1.0, initial_real_params=x0, sampling_width=0.05, adapt_shape=False,
               reject=lambda p: np.abs(p[:, 0]) >= 1.0)
    if shape == "shared":
        b = rng.standard_normal((nr, nr))
        cov = 0.5 * np.linalg.inv(0.5 * (amat + amat.T)) + 0.02 * (b @ b.T) / nr
        eng = me.MetropolisEngine(me.DenseQuadratic(amat), me.AbsReal0AtLeast(1.0), x0, None, cov_mode="pooled", **kw)
        eng.set_shared_factor(np.linalg.cholesky(cov)[np.tril_indices(nr)])
        ora = ManyChainOracle(nr, 0, energies.dense_quadratic(nr, 0, amat), n, covariance_matrix_real=cov, **okw)
    else:
        eng = me.MetropolisEngine(me.DenseQuadratic(amat), me.AbsReal0AtLeast(1.0), x0, None, cov_mode="fixed", **kw)
        ora = ManyChainOracle(nr, 0, energies.dense_quadratic(nr, 0, amat), n, **okw)
    assert np.allclose(eng.energy_total, ora.energy, rtol=1e-12)
    for k in range(16):
        sweeps = 1 if k % 2 else 3
        eng.step_all(sweeps)
        ora.step(sweeps)
        eng.measure()
        ora.measure()
    _compare(eng, ora, nr, 0)
    assert 0.05 < ora.accepted / ora.proposed < 0.95
    assert np.all(np.abs(eng._get(0)[:, 0]) < 1.0)


@pytest.mark.parametrize("nr,nc,dtype", [(96, 0, "f64"), (120, 0, "f64"), (120, 0, "f32"), (60, 25, "f64"), (140, 0, "f64"), (250, 0, "f32")])
def test_pooled_moments_and_adapt_pooled_shape_beyond_75_degrees_of_freedom(nr, nc, dtype):
    """me_pooled_moments used to stop at 3 072 moment entries (about 75 real degrees of freedom): stage 1 of the reduction
    now runs in passes of 3 072 entries (tiles of 64 / 32 / 16 chains by LDS size), so every engine that accepts
    cov_mode="pooled" can obtain its shared shape in-library.  Sums against numpy on the engine's own state."""
    from metropolisengine_amd import distributed
    n = 1000 + 37
    d = nr + 2 * nc
    a_r = np.linspace(0.5, 2.0, nr)
    a_c = np.linspace(0.7, 1.5, nc)
    pure_real = nc == 0
    eng = me.MetropolisEngine(me.DiagQuadratic(a_r, a_c), None, list(np.linspace(-0.2, 0.2, nr)),
                              (list(0.1 * np.exp(1j * np.arange(nc))) if nc else None), temp=1.0, n_chains=n, seed=4, dtype=dtype,
                              sampling_width=0.1, cov_mode="pooled" if pure_real else "fixed")
    eng.step_all(40)
    x = eng._get(0)
    got = eng.pooled_moments()
    assert got.shape == (distributed.moments_size(nr, nc),)
    tol = dict(rtol=1e-12, atol=1e-9) if dtype == "f64" else dict(rtol=2e-5, atol=2e-3)
    assert got[0] == n
    assert np.allclose(got[1:1 + d], x.sum(axis=0), **tol)
    second = x.T @ x
    assert np.allclose(got[1 + d:1 + d + d * (d + 1) // 2], second[np.tril_indices(d)], **tol)
    z = x[:, nr:nr + nc] + 1j * x[:, nr + nc:]
    obs = np.concatenate((np.abs(x[:, :nr]).sum(axis=0), np.abs(z).sum(axis=0), (x[:, :nr] ** 2).sum(axis=0)))
    assert np.allclose(got[1 + d + d * (d + 1) // 2:-2], obs, **tol)
    assert (got[-2], got[-1]) == tuple(float(v) for v in eng.accept_stats())
    if pure_real:
        stats = distributed.adapt_pooled_shape(eng, jitter=1e-6)      # pooled covariance -> Cholesky -> shared factor
        assert np.allclose(stats["covariance"], np.cov(x.T, bias=True), **tol)
        assert np.allclose(eng.shared_factor(), distributed.pooled_factor(stats["covariance"], nr, 0, jitter=1e-6))
        eng.step_all(3)                                               # and the engine steps with it
        assert 0.0 < eng.acceptance_rate() < 1.0


def test_unsupported_combinations_fail_loudly():
    with pytest.raises(NotImplementedError, match="identity proposal shape"):          # one shared factor with complex parameters
        me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 100, [0j] * 30, temp=1.0, n_chains=8, cov_mode="pooled")
    with pytest.raises(NotImplementedError, match="LDS"):                              # per-chain shapes: D x 64 values of LDS
        me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 400, None, temp=1.0, n_chains=8, dtype="f64")
    with pytest.raises(NotImplementedError, match="LDS"):
        me.MetropolisEngine(me.DenseQuadratic(np.identity(700)), None, [0.0] * 700, None, temp=1.0, n_chains=8,
                            cov_mode="fixed")                                                  # x' would not fit in LDS
    with pytest.raises(NotImplementedError):
        me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 100, [0j] * 10, temp=1.0, n_chains=8, cov_mode="pooled")
    # a non-identity initial covariance makes the engine start with a SHARED factor: refused at creation (not at the first
    # step) where the runtime set has no shared-factor form -- complex parameters, or 2 x D x 64 values beyond the LDS
    with pytest.raises(NotImplementedError, match="identity proposal shape only"):
        me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 100, [0j] * 10, covariance_matrix_real=2.0 * np.identity(100),
                            temp=1.0, n_chains=8, cov_mode="fixed")
    with pytest.raises(NotImplementedError, match="LDS"):
        me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 200, None, covariance_matrix_real=2.0 * np.identity(200),
                            temp=1.0, n_chains=8, cov_mode="fixed", dtype="f64")
    eng = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 120, None, temp=1.0, n_chains=8, cov_mode="fixed")
    with pytest.raises(NotImplementedError):
        eng.covariance_matrix_real


@pytest.mark.parametrize("dtype,energy_kind", [("f64", "diag"), ("f64", "dense"), ("f32", "diag")])
def test_per_chain_shapes_at_runtime_dimensions_follow_the_oracle(dtype, energy_kind):
    """140 real parameters with the reference's default semantics (cov_mode="reference"): beyond build.MAX_COMPILED_DOF the
    runtime-dimension set keeps the per-chain covariance (k_measure_runtime_cov), refreshes the per-chain factors
    (k_factor_runtime) and draws x' = x + sigma L_chain g column by column (k_step_runtime_lds<CK_PER_CHAIN>) --
    metropolis_engine.py:416-421 feeding :261-272 -- across the 50-measure threshold."""
    nr, n, seed = 140, 70, 61
    assert nr > build.MAX_COMPILED_DOF
    x0 = list(np.linspace(-0.2, 0.2, nr))
    if energy_kind == "diag":
        weights = tuple(np.linspace(0.5, 2.0, nr))
        energy, oracle_energy = me.DiagQuadratic(weights), energies.diag_quadratic(nr, 0, weights, ())
    else:
        m = np.random.default_rng(8).standard_normal((nr, nr))
        amat = m @ m.T / nr + np.identity(nr)
        energy, oracle_energy = me.DenseQuadratic(amat), energies.dense_quadratic(nr, 0, amat)
    eng = me.MetropolisEngine(energy, None, x0, None, temp=1.0, n_chains=n, seed=seed, dtype=dtype, sampling_width=0.1)
    assert eng.cov_mode == "reference"
    if dtype == "f32":
        # float32: the kernels run and the adaptive shapes are live (statistics, no trajectory parity at this length)
        for k in range(56):
            eng.step_all(2)
            eng.measure()
        fr, _ = eng.proposal_factors()
        cov = eng.covariance_matrix_real
        assert np.all(np.isfinite(fr)) and np.all(np.isfinite(cov))
        assert np.allclose(np.einsum("cij,ckj->cik", fr, fr), cov, rtol=2e-3, atol=2e-5)      # factor = chol(cov), per chain
        before = eng.accept_stats()
        eng.step_all(5)
        assert eng.accept_stats()[0] > before[0]
        return
    ora = ManyChainOracle(nr, 0, oracle_energy, n, seed=seed, temp=1.0, initial_real_params=x0, sampling_width=0.1)
    for k in range(54):
        eng.step_all(2)
        ora.step(2)
        eng.measure()
        ora.measure()
    fr, _ = eng.proposal_factors()
    assert np.allclose(eng.real_mean, ora.mean[:, :nr], rtol=0, atol=1e-10)
    assert np.allclose(eng.covariance_matrix_real, ora.cov_real, rtol=0, atol=1e-9)
    assert np.allclose(fr, ora.factor_real, rtol=0, atol=1e-8)
    for sweeps in (1, 3):
        eng.step_all(sweeps)
        ora.step(sweeps)
    assert np.allclose(eng._get(0), ora.x, rtol=0, atol=1e-8)
    assert np.allclose(eng.energy_total, ora.energy, rtol=0, atol=1e-8)
    assert eng.accept_stats() == (ora.accepted, ora.proposed)
    eng.measure()
    ora.measure()
    assert np.allclose(eng.covariance_matrix_real, ora.cov_real, rtol=0, atol=1e-9)


def test_tracked_covariance_of_a_mixed_space_at_runtime_dimensions():
    """100 real + 12 complex parameters (124 degrees of freedom), identity shape, track_covariance=True: the runtime-dimension
    measure kernel keeps each chain's running covariance, real block and Hermitian block (metropolis_engine.py:416-427), as
    statistics -- against the oracle across the 50-measure threshold."""
    nr, nc, n, seed = 100, 12, 70, 13
    real_w, cplx_w = tuple(np.linspace(0.5, 1.5, nr)), tuple(np.linspace(0.8, 1.2, nc))
    x0, z0 = list(np.linspace(-0.1, 0.1, nr)), list(0.05 * np.exp(1j * np.arange(nc)))
    eng = me.MetropolisEngine(me.DiagQuadratic(real_w, cplx_w), None, x0, z0, temp=1.0, n_chains=n, seed=seed, dtype="f64",
                              sampling_width=0.05, cov_mode="fixed", track_covariance=True)
    ora = ManyChainOracle(nr, nc, energies.diag_quadratic(nr, nc, real_w, cplx_w), n, seed=seed, temp=1.0,
                          initial_real_params=x0, initial_complex_params=z0, sampling_width=0.05, adapt_shape=False)
    for k in range(55):
        eng.step_all(2)
        ora.step(2)
        eng.measure()
        ora.measure()
    assert np.allclose(eng.covariance_matrix_real, ora.cov_real, rtol=0, atol=1e-10)
    assert np.allclose(eng.covariance_matrix_complex, ora.cov_complex, rtol=0, atol=1e-10)
    assert not np.allclose(eng.covariance_matrix_real[0], np.identity(nr), atol=1e-4)


@pytest.mark.parametrize("nr,nc", [(100, 20), (0, 70)], ids=["mixed", "complex"])
def test_per_chain_shapes_of_mixed_and_complex_spaces_at_runtime_dimensions(nr, nc):
    """140 degrees of freedom with complex parameters, cov_mode="reference": the Hermitian block's running covariance
    (:423-427), L = chol(conj K) (quirk Q3, :292-298, k_factor_runtime_complex) and z' = z + (sigma_c / sqrt 2) L zeta column by
    column (:274-302) -- step_all and, for the mixed space, group-wise steps -- against the oracle across the threshold."""
    n, seed = 70, 71
    assert nr + 2 * nc > build.MAX_COMPILED_DOF
    real_w, cplx_w = tuple(np.linspace(0.5, 2.0, nr)), tuple(np.linspace(0.7, 1.6, nc))
    x0 = list(np.linspace(-0.2, 0.2, nr)) if nr else None
    z0 = list(0.1 * np.exp(0.7j * np.arange(nc)))
    eng = me.MetropolisEngine(me.DiagQuadratic(real_w, cplx_w), None, x0, z0, temp=1.0, n_chains=n, seed=seed, dtype="f64",
                              sampling_width=0.08)
    ora = ManyChainOracle(nr, nc, energies.diag_quadratic(nr, nc, real_w, cplx_w), n, seed=seed, temp=1.0,
                          initial_real_params=x0, initial_complex_params=z0, sampling_width=0.08)
    assert eng.cov_mode == "reference"
    for k in range(54):
        eng.step_all(2)
        ora.step(2)
        eng.measure()
        ora.measure()
    fr, fc = eng.proposal_factors()
    assert np.allclose(eng.covariance_matrix_complex, ora.cov_complex, rtol=0, atol=1e-9)
    assert np.allclose(fc, ora.factor_complex, rtol=0, atol=1e-8)
    if nr:
        assert np.allclose(eng.covariance_matrix_real, ora.cov_real, rtol=0, atol=1e-9)
        assert np.allclose(fr, ora.factor_real, rtol=0, atol=1e-8)
    for sweeps in (1, 3):
        eng.step_all(sweeps)
        ora.step(sweeps)
    assert np.allclose(eng._get(0), ora.x, rtol=0, atol=1e-8)
    assert eng.accept_stats() == (ora.accepted, ora.proposed)
    if nr:
        for group in ("real", "complex", "real"):
            getattr(eng, "step_%s_group" % group)()
            ora.step(1, group=group)
        assert np.allclose(eng._get(0), ora.x, rtol=0, atol=1e-8)
        assert eng.accept_stats() == (ora.accepted, ora.proposed)
    eng.measure()
    ora.measure()
    assert np.allclose(eng.covariance_matrix_complex, ora.cov_complex, rtol=0, atol=1e-9)


def test_runtime_per_chain_fields_beyond_4_gib_are_addressed_correctly():
    """140 real parameters x 2^16 chains in float64: each packed per-chain field is 9 870 x 8 B x 65 536 = 5.2 GB.  The last chains
    of the big engine equal a small engine that owns the same global chain ids -- covariance recursion, factor refresh and the
    column-wise proposal all walk the fields with 64-bit pointers (same Philox streams, same arithmetic: bitwise)."""
    nr, n = 140, 1 << 16
    kw = dict(temp=1.0, seed=17, sampling_width=0.08, dtype="f64")
    weights = tuple(np.linspace(0.5, 2.0, nr))
    lo, cnt = n - 200, 200
    big = me.MetropolisEngine(me.DiagQuadratic(weights), None, [0.0] * nr, None, n_chains=n, **kw)
    small = me.MetropolisEngine(me.DiagQuadratic(weights), None, [0.0] * nr, None, n_chains=cnt, chain_offset=lo, **kw)
    for eng in (big, small):
        for k in range(52):
            eng.step_all(1)
            eng.measure()
        eng.step_all(2)
    assert np.array_equal(big._get(0)[lo:], small._get(0))
    assert np.array_equal(big.energy_total[lo:], small.energy_total)
    assert np.array_equal(big.real_group_sampling_width[lo:], small.real_group_sampling_width)


def test_initial_covariance_with_per_chain_shapes_at_runtime_dimensions():
    """cov_mode="reference" with covariance_matrix_real / _complex given (the reference's warm start, metropolis_engine.py:17-24,
    :62-75) on 100 real + 20 complex parameters: the runtime-dimension set has no shared-factor form for such a space, the engine
    proposes from the per-chain factor field -- every chain's copy holds chol of the initial matrices -- from the first step on,
    and switches to the chains' own running covariances after 50 measures like any other engine."""
    nr, nc, n, seed = 100, 20, 70, 81
    rng = np.random.default_rng(2)
    a = rng.standard_normal((nr, nr))
    cov_r = 0.5 * np.identity(nr) + 0.3 * a @ a.T / nr
    b = rng.standard_normal((nc, nc)) + 1j * rng.standard_normal((nc, nc))
    cov_c = 0.8 * np.identity(nc) + 0.2 * b @ b.conj().T / nc
    real_w, cplx_w = tuple(np.linspace(0.5, 2.0, nr)), tuple(np.linspace(0.7, 1.6, nc))
    x0, z0 = list(np.linspace(-0.2, 0.2, nr)), list(0.1 * np.exp(0.7j * np.arange(nc)))
    eng = me.MetropolisEngine(me.DiagQuadratic(real_w, cplx_w), None, x0, z0, temp=1.0, n_chains=n, seed=seed, dtype="f64",
                              sampling_width=0.05, covariance_matrix_real=cov_r, covariance_matrix_complex=cov_c)
    ora = ManyChainOracle(nr, nc, energies.diag_quadratic(nr, nc, real_w, cplx_w), n, seed=seed, temp=1.0,
                          initial_real_params=x0, initial_complex_params=z0, sampling_width=0.05,
                          covariance_matrix_real=cov_r, covariance_matrix_complex=cov_c)
    eng.step_all(3)
    ora.step(3)
    assert np.allclose(eng._get(0), ora.x, rtol=0, atol=1e-9)
    assert eng.accept_stats() == (ora.accepted, ora.proposed)
    eng.step_real_group()
    ora.step(1, group="real")
    for k in range(52):
        eng.step_all(1)
        ora.step(1)
        eng.measure()
        ora.measure()
    eng.step_all(2)
    ora.step(2)
    assert np.allclose(eng._get(0), ora.x, rtol=0, atol=1e-8)
    assert np.allclose(eng.covariance_matrix_complex, ora.cov_complex, rtol=0, atol=1e-9)
    assert eng.accept_stats() == (ora.accepted, ora.proposed)


def test_checkpoint_round_trip_with_per_chain_shapes_at_runtime_dimensions():
    """state_dict / load_state_dict of a 100 real + 20 complex engine in cov_mode="reference" (per-chain covariance and factor
    fields of the runtime-dimension set included): the resumed engine continues the uninterrupted trajectory bit for bit, both
    when the checkpoint is taken before the 50-measure threshold and after it."""
    nr, nc = 100, 20
    args = (me.DiagQuadratic(tuple(np.linspace(0.5, 2.0, nr)), tuple(np.linspace(0.7, 1.6, nc))), None, [0.05] * nr, [0.05j] * nc)
    kw = dict(temp=1.0, n_chains=130, seed=19, dtype="f64", sampling_width=0.06)
    for measures_before in (20, 53):
        a = me.MetropolisEngine(*args, **kw)
        for _ in range(measures_before):
            a.step_all(1)
            a.measure()
        state = a.state_dict()
        b = me.MetropolisEngine(*args, **kw)
        b.load_state_dict(state)
        for eng in (a, b):
            for _ in range(35 if measures_before < 50 else 3):      # the first engine crosses the threshold after the restart
                eng.step_all(2)
                eng.measure()
            eng.step_complex_group()
        for field in range(7):
            assert np.array_equal(a._get(field), b._get(field)), (measures_before, field)
        assert a.accept_stats() == b.accept_stats()


def test_cycle_on_the_runtime_dimension_set_is_step_all_plus_measure():
    """engine.cycle(k) where no one-launch kernel exists (the runtime-dimension set): me_cycle issues the step launch and the
    measure launch itself -- same state, statistics and per-chain shapes as step_all(k); measure(), across the threshold."""
    nr = 140
    args = (me.DiagQuadratic(tuple(np.linspace(0.5, 2.0, nr))), None, [0.02] * nr, None)
    kw = dict(temp=1.0, n_chains=100, seed=23, dtype="f64", sampling_width=0.06)
    a, b = me.MetropolisEngine(*args, **kw), me.MetropolisEngine(*args, **kw)
    for _ in range(54):
        a.cycle(2)
        b.step_all(2)
        b.measure()
    assert a.fused_cycles() == 0
    for field in range(7):
        assert np.array_equal(a._get(field), b._get(field)), field
    assert a.accept_stats() == b.accept_stats()
