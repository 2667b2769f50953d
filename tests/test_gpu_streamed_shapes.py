"""Per-chain adaptive proposal shapes beyond the register-resident size (more than 160 packed matrix entries): BASELINE
config 4's parameter space -- 64 real parameters, 2 080 entries per chain -- in cov_mode="reference", i.e. with the
reference's own semantics: every chain proposes with the Cholesky factor of ITS running covariance once more than 50
measures have been taken (metropolis_engine.py:389, :416-421 feeding :268-270).  The covariance is streamed by k_measure,
the factor rebuilt by k_factor_stream, the proposals x' = x + sigma L g stream the factor in k_step (csrc/me_device.h).
float64 follows the many-chain oracle (numpy Cholesky per chain) on the same Philox streams."""
import numpy as np
import pytest

import metropolisengine_amd as me
from metropolisengine_amd.engine import unpack_real_block, unpack_real_factor
from oracle import energies
from oracle.manychain import ManyChainOracle

pytestmark = pytest.mark.gpu

_M = np.random.default_rng(5).standard_normal((64, 64))
AMAT = _M @ _M.T / 64 + np.identity(64)


@pytest.mark.parametrize("kind", ["dense", "iso"])
def test_64_real_per_chain_shapes_follow_the_oracle(kind):
    n, seed = 70, 41                                  # ragged: one full wavefront tile and six chains
    x0 = list(np.linspace(-0.2, 0.2, 64))
    if kind == "dense":
        spec, energy = me.DenseQuadratic(AMAT), energies.dense_quadratic(64, 0, AMAT)
    else:
        spec, energy = me.IsoQuadratic(0.8), energies.iso_quadratic(64, 0, 0.8)
    eng = me.MetropolisEngine(spec, None, x0, None, temp=1.0, n_chains=n, seed=seed, dtype="f64", sampling_width=0.1)
    ora = ManyChainOracle(64, 0, energy, n, seed=seed, temp=1.0, initial_real_params=x0, sampling_width=0.1)
    assert eng.cov_mode == "reference"
    for k in range(58):                               # the shape switches on after the 50th measure
        eng.step_all(2)
        ora.step(2)
        eng.measure()
        ora.measure()
    assert eng.measure_step_counter == ora.measure_step_counter == 59
    cov = eng.covariance_matrix_real
    assert cov.shape == (n, 64, 64)
    assert np.allclose(cov, ora.cov_real, rtol=0, atol=1e-9)
    fr, _ = eng.proposal_factors()
    assert np.allclose(fr, ora.factor_real, rtol=0, atol=1e-8)
    assert np.allclose(fr @ np.swapaxes(fr, 1, 2), ora.cov_real, rtol=0, atol=1e-8)      # L L^T = C, chain by chain
    # ... and the proposals use it: one-sweep launches and fused sweeps with the streamed factor
    for sweeps in (1, 1, 3, 5):
        eng.step_all(sweeps)
        ora.step(sweeps)
    assert np.allclose(eng._get(0), ora.x, rtol=0, atol=1e-8)
    assert np.allclose(eng.energy_total, ora.energy, rtol=0, atol=1e-8)
    assert np.allclose(eng.real_group_sampling_width, ora.width_real, rtol=1e-11)
    assert eng.accept_stats() == (ora.accepted, ora.proposed)
    eng.measure()
    ora.measure()
    assert np.allclose(eng.covariance_matrix_real, ora.cov_real, rtol=0, atol=1e-8)


@pytest.mark.parametrize("nr", [24, 96])
def test_other_streamed_sizes_follow_the_oracle(nr):
    """The streamed kernels at the sizes either side of 64: 24 parameters (300 packed entries; k_factor_tile with TWO chains
    per wavefront, pivots by ds_bpermute) and 96 (4 656 entries; beyond a lane per row, so k_factor_stream), both compiled
    on demand (build.build_dims), float64 against the many-chain oracle."""
    n, seed = 75, 47
    x0 = list(np.linspace(-0.2, 0.2, nr))
    weights = tuple(np.linspace(0.5, 2.0, nr))
    eng = me.MetropolisEngine(me.DiagQuadratic(weights), None, x0, None, temp=1.0, n_chains=n, seed=seed, dtype="f64",
                              sampling_width=0.1, cov_mode="reference")
    ora = ManyChainOracle(nr, 0, energies.diag_quadratic(nr, 0, weights, ()), n, seed=seed, temp=1.0,
                          initial_real_params=x0, sampling_width=0.1)
    for k in range(54):
        eng.step_all(2)
        ora.step(2)
        eng.measure()
        ora.measure()
    fr, _ = eng.proposal_factors()
    assert np.allclose(eng.covariance_matrix_real, ora.cov_real, rtol=0, atol=1e-9)
    assert np.allclose(fr, ora.factor_real, rtol=0, atol=1e-8)
    for sweeps in (1, 3):
        eng.step_all(sweeps)
        ora.step(sweeps)
    assert np.allclose(eng._get(0), ora.x, rtol=0, atol=1e-8)
    assert np.allclose(eng.energy_total, ora.energy, rtol=0, atol=1e-8)
    assert eng.accept_stats() == (ora.accepted, ora.proposed)
    # the float32 build of the same kernels: its factors reproduce its covariance, chain by chain
    f32 = me.MetropolisEngine(me.DiagQuadratic(weights), None, x0, None, temp=1.0, n_chains=n, seed=seed, dtype="f32",
                              sampling_width=0.1, cov_mode="reference")
    for k in range(54):
        f32.step_all(2)
        f32.measure()
    f32.step_all(3)
    fr, _ = f32.proposal_factors()
    assert np.allclose(fr @ np.swapaxes(fr, 1, 2), f32.covariance_matrix_real, rtol=2e-4, atol=2e-5)
    assert np.allclose(f32.covariance_matrix_real, eng.covariance_matrix_real, rtol=0, atol=5e-3)   # same streams, near-equal paths


def test_64_real_per_chain_shapes_float32_shards_and_state():
    """float32: a sub-range of a larger engine is bitwise the small engine that owns the same chain ids (covariance and
    factor included), the factors reproduce their covariance, and a checkpoint resumes bit for bit."""
    kw = dict(temp=1.0, seed=43, sampling_width=0.1)
    x0 = [0.0] * 64
    big = me.MetropolisEngine(me.DenseQuadratic(AMAT), None, x0, None, n_chains=64 * 5 + 9, **kw)
    lo, cnt = 64 * 2 + 30, 100
    small = me.MetropolisEngine(me.DenseQuadratic(AMAT), None, x0, None, n_chains=cnt, chain_offset=lo, **kw)
    for eng in (big, small):
        for _ in range(54):
            eng.step_all(2)
            eng.measure()
        eng.step_all(4)
    for field in range(7):
        assert np.array_equal(big._get(field, lo, cnt), small._get(field)), field
    fr, _ = small.proposal_factors()
    assert np.allclose(fr @ np.swapaxes(fr, 1, 2), small.covariance_matrix_real, rtol=2e-4, atol=2e-5)
    clone = me.MetropolisEngine(me.DenseQuadratic(AMAT), None, x0, None, n_chains=cnt, chain_offset=lo, **kw)
    clone.load_state_dict(small.state_dict())
    for eng in (small, clone):
        eng.step_all(3)
        eng.measure()
        eng.step_all()
    for field in range(7):
        assert np.array_equal(small._get(field), clone._get(field)), field


def test_fixed_and_pooled_shapes_at_64_parameters_keep_no_per_chain_matrices():
    """The matrices cost 8-16 KB per chain: only cov_mode="reference" (or track_covariance) allocates them."""
    eng = me.MetropolisEngine(me.DenseQuadratic(AMAT), None, [0.0] * 64, None, temp=1.0, n_chains=64, seed=1, cov_mode="fixed")
    eng.step_all(5)
    eng.measure()
    with pytest.raises(NotImplementedError):
        eng.covariance_matrix_real


@pytest.mark.parametrize("nr,nc", [(1, 13), (0, 13)])
def test_mixed_and_complex_spaces_beyond_160_entries_follow_the_oracle(nr, nc):
    """Round 3: per-chain adaptive shapes for spaces WITH a complex block beyond the register-resident size (1 real + 13
    complex = 170 packed entries; 13 complex = 169): the Hermitian covariance streams through k_measure, k_factor_mixed builds
    chol(C_r) and chol(conj K) (quirk Q3, metropolis_engine.py:292-298), and k_step streams the complex rows of the factor
    behind the real triangle (:274-302).  float64 against the many-chain oracle across the 50-measure threshold; the
    group-wise steps of the mixed space stream the same factor."""
    n, seed = 70, 43
    a, b = tuple([0.7] * nr), tuple(0.5 + 0.25 * j for j in range(nc))
    real0 = [0.1] * nr if nr else None
    cplx0 = [0.05 * (j + 1) * np.exp(0.3j * j) for j in range(nc)]
    eng = me.MetropolisEngine(me.DiagQuadratic(a, b), None, real0, cplx0, temp=1.0, n_chains=n, seed=seed, dtype="f64",
                              sampling_width=0.2)
    ora = ManyChainOracle(nr, nc, energies.diag_quadratic(nr, nc, a, b), n, seed=seed, temp=1.0, initial_real_params=real0,
                          initial_complex_params=cplx0, sampling_width=0.2)
    assert eng.cov_mode == "reference"
    for _ in range(56):
        eng.step_all(3)
        ora.step(3)
        eng.measure()
        ora.measure()
    assert np.abs(ora.cov_complex[0] - np.identity(nc)).max() > 1e-3          # the shapes really left the identity
    assert np.allclose(eng.covariance_matrix_complex, ora.cov_complex, rtol=0, atol=1e-9)
    fr, fc = eng.proposal_factors()
    assert np.allclose(fc, ora.factor_complex, rtol=0, atol=1e-8)
    assert np.allclose(fc @ np.conj(np.swapaxes(fc, 1, 2)), np.conj(ora.cov_complex), rtol=0, atol=1e-8)   # L L^H = conj K
    if nr:
        assert np.allclose(fr, ora.factor_real, rtol=0, atol=1e-8)
    for sweeps in (1, 1, 3, 5):
        eng.step_all(sweeps)
        ora.step(sweeps)
    assert np.allclose(eng._get(0), ora.x, rtol=0, atol=1e-8)
    assert np.allclose(eng.energy_total, ora.energy, rtol=0, atol=1e-8)
    assert eng.accept_stats() == (ora.accepted, ora.proposed)
    if nr:                                          # group-wise steps of the mixed space
        eng.step_real_group(2)
        ora.step(2, group="real")
        eng.step_complex_group(2)
        ora.step(2, group="complex")
        assert np.allclose(eng._get(0), ora.x, rtol=0, atol=1e-8)
        assert eng.accept_stats() == (ora.accepted, ora.proposed)
    eng.measure()
    ora.measure()
    assert np.allclose(eng.covariance_matrix_complex, ora.cov_complex, rtol=0, atol=1e-8)


def test_per_chain_shapes_beyond_96_degrees_of_freedom_built_in_energy():
    """100 real parameters, built-in energy, the reference's default semantics (cov_mode="reference"): beyond 96 degrees of
    freedom the runtime-dimension set has no per-chain shapes, so the constructor compiles the space's own kernel set (up to
    build.MAX_COMPILED_DOF = 128; prebuilt for this test by build_examples) and the streamed-shape kernels take over."""
    nr, n, seed = 100, 70, 51
    x0 = list(np.linspace(-0.2, 0.2, nr))
    weights = tuple(np.linspace(0.5, 2.0, nr))
    eng = me.MetropolisEngine(me.DiagQuadratic(weights), None, x0, None, temp=1.0, n_chains=n, seed=seed, dtype="f64",
                              sampling_width=0.1)
    ora = ManyChainOracle(nr, 0, energies.diag_quadratic(nr, 0, weights, ()), n, seed=seed, temp=1.0,
                          initial_real_params=x0, sampling_width=0.1)
    assert eng.cov_mode == "reference"
    for k in range(54):
        eng.step_all(2)
        ora.step(2)
        eng.measure()
        ora.measure()
    fr, _ = eng.proposal_factors()
    assert np.allclose(eng.covariance_matrix_real, ora.cov_real, rtol=0, atol=1e-9)
    assert np.allclose(fr, ora.factor_real, rtol=0, atol=1e-8)
    for sweeps in (1, 3):
        eng.step_all(sweeps)
        ora.step(sweeps)
    assert np.allclose(eng._get(0), ora.x, rtol=0, atol=1e-8)
