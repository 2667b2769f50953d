"""GPU parity tests: the HIP path (through the ctypes C-ABI) against the CPU oracle and the golden fixtures.

Tolerances
 * float64 engines vs the float64 oracle on identical streams: 1e-9 absolute on O(1) quantities.  Both sides do the
   same arithmetic; they differ by libm vs device transcendentals (<= 2 ulp), FMA contraction and the one-pass
   covariance form, so agreement is expected at ~1e-13 and every accept decision must coincide.
 * float32 engines: compared statistically (pooled moments vs the analytic stationary values T/(2a), T/b within
   4-5 standard errors) and on a one-step horizon against the oracle (1e-5).
"""
import os

import numpy as np
import pytest

import metropolisengine_amd as me
from oracle import energies, scenarios
from oracle.manychain import ManyChainOracle

pytestmark = pytest.mark.gpu
TOL = 1e-9

_rng = np.random.default_rng(5)
_M = _rng.standard_normal((6, 6))
DENSE6 = _M @ _M.T / 6 + np.identity(6)

# name -> (nr, nc, product spec, oracle energy, temp, real0, cplx0, steps/measure, measures, product reject, oracle reject)
CASES = {
    "iso_16real": (16, 0, me.IsoQuadratic(1.0), energies.iso_quadratic(16, 0, 1.0), 1.0, [0.0] * 16, None, 2, 70,
                   None, None),
    "diag_4real_4complex": (4, 4, me.DiagQuadratic((1, 2, 4, 8), (1, 2, 4, 8)),
                            energies.diag_quadratic(4, 4, (1, 2, 4, 8), (1, 2, 4, 8)), 1.0,
                            [0.1, 0.2, -0.1, 0.0], [0.1j, 0.2, -0.1 + 0.1j, 0.0], 3, 80, None, None),
    "dense_2real_2complex": (2, 2, me.DenseQuadratic(DENSE6), energies.dense_quadratic(2, 2, DENSE6), 0.7,
                             [0.3, -0.3], [0.1 + 0.1j, -0.2j], 2, 90, None, None),
    "diag_3complex": (0, 3, me.DiagQuadratic((), (1, 3, 0.5)), energies.diag_quadratic(0, 3, (), (1, 3, 0.5)), 0.5,
                      None, [0.1, 0.1j, -0.1], 2, 70, None, None),
    "landau": (2, 1, me.LandauToy(), energies.landau_toy(), 0.1, [0.0, 0.0], [0j], 10, 60, None, None),
    "wall": (1, 1, me.DiagQuadratic((0.5,), (1.0,)), energies.diag_quadratic(1, 1, (0.5,), (1.0,)), 1.0, [0.0], [0.1j],
             3, 70, me.AbsReal0AtLeast(0.25), energies.wall_reject(0.25)),
    "zero_temp": (2, 0, me.DiagQuadratic((1.0, 3.0)), energies.diag_quadratic(2, 0, (1.0, 3.0)), 0.0, [1.0, -1.0], None,
                  4, 30, None, None),
    "cylinder_2real_7complex": (2, 7, me.CylinderSurrogate(1.0, 0.5, 1.0), energies.cylinder_surrogate(2, 7, 1.0, 0.5, 1.0),
                                0.1, [0.1, 0.0], [0.05] * 7, 5, 60, me.AbsReal0AtLeast(1.0),
                                energies.wall_reject(1.0)),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_f64_trajectories_follow_the_oracle(name):
    nr, nc, spec, oracle_energy, temp, real0, cplx0, spm, nm, rej, oracle_rej = CASES[name]
    n_chains, seed, offset = 192, 2026, (1 << 33) + 17
    eng = me.MetropolisEngine(spec, rej, real0, cplx0, temp=temp, n_chains=n_chains, seed=seed, dtype="f64",
                              chain_offset=offset)
    ora = ManyChainOracle(nr, nc, oracle_energy, n_chains, seed=seed, temp=temp, initial_real_params=real0,
                          initial_complex_params=cplx0, chain_offset=offset, reject=oracle_rej)
    assert abs(eng.alpha - ora.alpha) < 1e-12 and abs(eng.ratio - ora.ratio) < 1e-12 and eng.m == ora.m
    assert np.allclose(eng.energy_total, ora.energy, rtol=0, atol=TOL)
    for k in range(nm):
        eng.step_all(spm)
        ora.step(spm)
        eng.measure()
        ora.measure()
        if k % 10 == 9 or k == nm - 1:
            x = eng._get(0)
            assert np.allclose(x, ora.x, rtol=0, atol=TOL), "state differs after measure %d" % k
            width = eng._get(2)[:, 0]
            assert np.allclose(width, ora.width_real if nr else ora.width_complex, rtol=0, atol=TOL)
            assert np.allclose(eng._get(3), ora.mean, rtol=0, atol=TOL)
            assert np.allclose(eng._get(5), ora.observables_mean, rtol=0, atol=TOL)
            if nr:
                assert np.allclose(eng.covariance_matrix_real, ora.cov_real, rtol=0, atol=TOL)
            if nc:
                assert np.allclose(eng.covariance_matrix_complex, ora.cov_complex, rtol=0, atol=TOL)
            assert np.allclose(eng.energy_total, ora.energy, rtol=0, atol=TOL)
    fr, fc = eng.proposal_factors()
    if nr:
        assert np.allclose(fr, ora.factor_real, rtol=0, atol=1e-8)
    if nc:
        assert np.allclose(fc, ora.factor_complex, rtol=0, atol=1e-8)
    assert eng.accept_stats() == (ora.accepted, ora.proposed)
    assert eng.measure_step_counter == ora.measure_step_counter
    pooled = eng.pooled_moments()
    assert np.allclose(pooled, ora.pooled_moments(), rtol=1e-11, atol=1e-9)


# ---------------------------------------------------------------- golden fixtures replayed through the HIP kernels
_COUPLED = np.zeros((6, 6))
_COUPLED[0, 0], _COUPLED[1, 1] = 1.0, 2.0
_COUPLED[0, 1] = _COUPLED[1, 0] = 0.3
_COUPLED[2, 2] = _COUPLED[4, 4] = 1.5
_COUPLED[3, 3] = _COUPLED[5, 5] = 3.0
_COUPLED[2, 3] = _COUPLED[3, 2] = 0.4
_COUPLED[4, 5] = _COUPLED[5, 4] = 0.4
_COUPLED[4, 3] = _COUPLED[3, 4] = 0.7
_COUPLED[2, 5] = _COUPLED[5, 2] = -0.7

GOLDEN_SPECS = {
    "readme_1real": (me.IsoQuadratic(1.0), None),
    "well_2real": (me.IsoQuadratic(1.0), None),
    "aniso_3real": (me.DiagQuadratic((1.0, 2.0, 4.0)), None),
    "aniso_2complex": (me.DiagQuadratic((), (1.0, 3.0)), None),
    "coupled_2real_2complex": (me.DenseQuadratic(_COUPLED), None),
    "landau_toy": (me.LandauToy(1.0, -1.0, 0.5), None),
    "wall_1real_1complex": (me.DiagQuadratic((0.5,), (1.0,)), me.AbsReal0AtLeast(0.25)),
    "zero_temp_2real": (me.DiagQuadratic((1.0, 3.0)), None),
}


@pytest.mark.parametrize("name", sorted(GOLDEN_SPECS))
def test_reference_golden_trajectory_on_gpu(name, golden_dir):
    """The reference's own injected-stream trajectories (tests/golden/traj_*.npz, recorded from
    metropolisengine/metropolis_engine.py) replayed through k_step/k_measure in float64 via me_step_injected."""
    spec = scenarios.SCENARIOS[name]
    gold = np.load(os.path.join(golden_dir, "traj_%s.npz" % name))
    energy_spec, reject = GOLDEN_SPECS[name]
    nr, nc = scenarios.dims(spec)
    eng = me.MetropolisEngine(energy_spec, reject, spec["real"], spec["cplx"], temp=spec["temp"], n_chains=1,
                              dtype="f64")
    assert np.allclose([eng.alpha, eng.m, eng.ratio], gold["constants"], rtol=0, atol=1e-12)
    spm = spec["steps_per_measure"]
    t = 0
    accepted = 0
    for k in range(spec["n_measures"]):
        eng.step_injected(gold["normals"][t:t + spm, None, :], gold["uniforms"][t:t + spm, :1])
        t += spm
        eng.measure()
        accepted_now = eng.accept_stats()[0]
        assert accepted_now == int(np.sum(gold["accept"][:t])), "accept decisions differ by step %d" % t
        accepted = accepted_now
        assert np.allclose(eng.real_params, gold["real_params"][t - 1], rtol=0, atol=TOL)
        assert np.allclose(eng.complex_params, gold["complex_params"][t - 1], rtol=0, atol=TOL)
        assert abs(eng.real_group_sampling_width - gold["real_width"][t - 1]) < TOL
        assert abs(eng.complex_group_sampling_width - gold["complex_width"][t - 1]) < TOL
        assert np.allclose(eng.real_mean, gold["real_mean"][k], rtol=0, atol=TOL)
        assert np.allclose(eng.complex_mean, gold["complex_mean"][k], rtol=0, atol=TOL)
        if nr:
            assert np.allclose(eng.covariance_matrix_real, gold["cov_real"][k], rtol=0, atol=TOL)
        if nc:
            assert np.allclose(eng.covariance_matrix_complex, gold["cov_complex"][k], rtol=0, atol=TOL)
        assert np.allclose(eng.observables_mean, gold["observables_mean"][k], rtol=0, atol=TOL)
    assert accepted == int(np.sum(gold["accept"]))


# ---------------------------------------------------------------- float32 (the production dtype)
def test_f32_one_step_matches_oracle():
    n, seed = 4096, 7
    eng = me.MetropolisEngine(me.DiagQuadratic((1, 2, 4, 8), (1, 2, 4, 8)), None, [0.1, 0.2, -0.1, 0.0],
                              [0.1j, 0.2, -0.1 + 0.1j, 0.0], temp=1.0, n_chains=n, seed=seed, dtype="f32")
    ora = ManyChainOracle(4, 4, energies.diag_quadratic(4, 4, (1, 2, 4, 8), (1, 2, 4, 8)), n, seed=seed, temp=1.0,
                          initial_real_params=[0.1, 0.2, -0.1, 0.0], initial_complex_params=[0.1j, 0.2, -0.1 + 0.1j, 0.0])
    eng.step_all()
    ora.step()
    x = eng._get(0)
    same = np.all(np.abs(x - ora.x) < 1e-5, axis=1)
    assert same.mean() > 0.999          # a float32 accept decision may flip only on a near-tie
    acc, prop = eng.accept_stats()
    assert prop == n and abs(acc - ora.accepted) <= 4


def test_f32_stationary_moments_16_real():
    """BASELINE config 2 shape at 2^16 chains: Var x_i = T/(2a) = 0.5, mean 0, acceptance near the 0.3 target."""
    n = 1 << 16
    eng = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 16, None, temp=1.0, n_chains=n, seed=2026,
                              sampling_width=0.3)
    eng.step_all(1500)
    stats_before = eng.accept_stats()
    eng.step_all(100)
    acc, prop = eng.accept_stats()
    rate = (acc - stats_before[0]) / (prop - stats_before[1])
    assert 0.2 < rate < 0.4
    from metropolisengine_amd.distributed import moments_to_statistics
    st = moments_to_statistics(eng.pooled_moments(), 16, 0)
    se_var = 0.5 * np.sqrt(2.0 / n)
    assert np.all(np.abs(np.diag(st["covariance"]) - 0.5) < 5 * se_var)
    assert np.all(np.abs(st["mean"]) < 5 * np.sqrt(0.5 / n))
    off = st["covariance"] - np.diag(np.diag(st["covariance"]))
    assert np.all(np.abs(off) < 5 * 0.5 / np.sqrt(n))
    # observables: <|x|> = sqrt(T/(pi a)), <x^2> = T/(2a)   
    assert np.all(np.abs(st["observables_mean"][:16] - np.sqrt(1.0 / np.pi)) < 5 * 0.43 / np.sqrt(n))
    assert np.all(np.abs(st["observables_mean"][16:] - 0.5) < 5 * se_var)


def test_f32_stationary_moments_mixed_adaptive():
    """BASELINE config 3: 4 real + 4 complex anisotropic, adaptive covariance on (> 50 measures)."""
    n = 1 << 15
    a = b = (1.0, 2.0, 4.0, 8.0)
    eng = me.MetropolisEngine(me.DiagQuadratic(a, b), None, [0.0] * 4, [0j] * 4, temp=1.0, n_chains=n, seed=11,
                              sampling_width=0.3)
    for _ in range(120):
        eng.step_all(10)
        eng.measure()
    eng.sync()
    from metropolisengine_amd.distributed import moments_to_statistics
    st = moments_to_statistics(eng.pooled_moments(), 4, 4)
    var = np.diag(st["covariance"])
    want = np.concatenate((1.0 / (2 * np.array(a)), 1.0 / (2 * np.array(b)), 1.0 / (2 * np.array(b))))
    assert np.all(np.abs(var / want - 1) < 5 * np.sqrt(2.0 / n))
    # per-chain reference-style covariance converges to Sigma + ~sigma^2 I (quirk Q1): just require SPD + finite
    cov = eng.covariance_matrix_real
    assert np.all(np.isfinite(cov)) and np.all(np.linalg.eigvalsh(cov[:64]) > 0)
    assert 0.15 < eng.acceptance_rate() < 0.5


def test_fused_sweeps_equal_single_sweeps_bitwise():
    kw = dict(temp=1.0, n_chains=1000, seed=5)
    a = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 16, None, **kw)
    b = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 16, None, **kw)
    a.step_all(12)
    for _ in range(12):
        b.step_all()
    assert np.array_equal(a._get(0), b._get(0)) and np.array_equal(a._get(2), b._get(2))
    assert a.accept_stats() == b.accept_stats() == (a.accept_stats()[0], 12000)


def test_sharding_invariance_bitwise():
    """Streams are addressed by global chain id: two half shards reproduce the full engine bit for bit."""
    args = (me.LandauToy(), None, [0.0, 0.0], [0j])
    full = me.MetropolisEngine(*args, temp=0.1, n_chains=600, seed=3)
    lo = me.MetropolisEngine(*args, temp=0.1, n_chains=300, seed=3, chain_offset=0)
    hi = me.MetropolisEngine(*args, temp=0.1, n_chains=300, seed=3, chain_offset=300)
    for _ in range(60):
        for e in (full, lo, hi):
            e.step_all(5)
            e.measure()
    for field in range(7):
        assert np.array_equal(full._get(field), np.concatenate((lo._get(field), hi._get(field)))), field
    # float32 engines sum each 64-chain tile in float32 (fp64 across tiles): shards tile differently -> ~1e-7 relative
    assert np.allclose(full.pooled_moments(), lo.pooled_moments() + hi.pooled_moments(), rtol=5e-6, atol=1e-4)


def test_checkpoint_round_trip():
    args = (me.DiagQuadratic((1, 2, 4, 8), (1, 2, 4, 8)), None, [0.0] * 4, [0j] * 4)
    a = me.MetropolisEngine(*args, temp=1.0, n_chains=500, seed=9)
    for _ in range(55):
        a.step_all(3)
        a.measure()
    state = a.state_dict()
    b = me.MetropolisEngine(*args, temp=1.0, n_chains=500, seed=9)
    b.load_state_dict(state)
    for e in (a, b):
        for _ in range(5):
            e.step_all(3)
            e.measure()
    for field in range(7):
        assert np.array_equal(a._get(field), b._get(field)), field


def test_checkpoint_round_trip_pooled_shape_and_counters():
    """cov_mode="pooled": the shared proposal factor and the acceptance counters are part of the checkpoint; a resumed run
    continues the uninterrupted trajectory bit for bit."""
    from metropolisengine_amd.distributed import adapt_pooled_shape
    args = (me.DiagQuadratic((1, 2, 4, 8), (1, 2, 4, 8)), None, [0.0] * 4, [0j] * 4)
    kw = dict(temp=1.0, n_chains=700, seed=10, cov_mode="pooled")
    a = me.MetropolisEngine(*args, **kw)
    a.step_all(300)
    assert a.shared_factor() is None and "shared_factor" not in a.state_dict()
    adapt_pooled_shape(a)
    a.step_all(40)
    a.measure()
    state = a.state_dict()
    assert state["shared_factor"].shape == (4 * 5 // 2 + 16,) and state["proposed"] == 700 * 340
    b = me.MetropolisEngine(*args, **kw)
    b.load_state_dict(state)
    assert b.accept_stats() == a.accept_stats()
    for e in (a, b):
        e.step_all(25)
        e.measure()
    for field in range(5):
        assert np.array_equal(a._get(field), b._get(field)), field
    assert a.accept_stats() == b.accept_stats()
    assert np.array_equal(a.shared_factor(), b.shared_factor())


def test_checkpoint_round_trip_reference_energy_ledgers_and_validation():
    """Quirk-Q5 mode keeps a separate energy_total row: it travels with the checkpoint, and a checkpoint that does not fit
    the engine it is loaded into (other flags, other mode, other shapes) raises BEFORE anything is written."""
    args = (me.DiagQuadratic((1, 2), (1, 2)), None, [0.1, 0.2], [0.1j, 0.2])
    kw = dict(temp=1.0, n_chains=300, seed=12, dtype="f64")
    a = me.MetropolisEngine(*args, reference_energy_ledgers=True, **kw)
    for _ in range(6):
        a.step_all(2)
        a.step_real_group()
        a.step_complex_group()
        a.measure()
    state = a.state_dict()
    assert "energy_total" in state
    b = me.MetropolisEngine(*args, reference_energy_ledgers=True, **kw)
    b.load_state_dict(state)
    for e in (a, b):
        e.step_all(3)
        e.step_real_group(2)
        e.measure()
    for field in (0, 1, 2, 3, 4, 5, 7):
        assert np.array_equal(a._get(field), b._get(field)), field
    assert a.accept_stats() == b.accept_stats()
    # ... into an engine WITHOUT the two-ledger flag: refused, and the target keeps its state, counters and statistics
    plain = me.MetropolisEngine(*args, **kw)
    plain.step_all(4)
    before = [plain._get(f) for f in range(6)], plain.accept_stats(), plain.measure_step_counter, plain.step_counter
    with pytest.raises(NotImplementedError):
        plain.load_state_dict(state)
    wrong_shape = dict(plain.state_dict())
    wrong_shape["mean"] = wrong_shape["mean"][:100]
    with pytest.raises(ValueError, match="shape"):
        plain.load_state_dict(wrong_shape)
    pooled_state = dict(plain.state_dict(), shared_factor=np.ones(2 * 3 // 2 + 4))
    with pytest.raises(ValueError, match="pooled"):
        plain.load_state_dict(pooled_state)
    after = [plain._get(f) for f in range(6)], plain.accept_stats(), plain.measure_step_counter, plain.step_counter
    assert all(np.array_equal(x, y) for x, y in zip(before[0], after[0])) and before[1:] == after[1:]


def test_set_rejects_partial_mixed_widths_before_writing():
    """me_set validates before any device write: a partial width update of a mixed engine leaves the widths untouched."""
    eng = me.MetropolisEngine(me.DiagQuadratic((1, 2), (1, 2)), None, [0.0] * 2, [0j] * 2, temp=1.0, n_chains=64, seed=3)
    eng.step_all(20)
    before = eng._get(2)
    with pytest.raises(ValueError):
        eng._set(2, np.full((10, 3), 7.0), chain_begin=5)
    assert np.array_equal(eng._get(2), before)


def test_reference_api_surface_single_chain():
    """README.md:26-51 transcribed: one real parameter, E = x^2, T = 0.01, 1000 x (step_all, measure)."""
    eng = me.MetropolisEngine(me.IsoQuadratic(1.0), initial_real_params=[0.0], temp=.01, seed=12345, dtype="f64")
    accepts = 0
    for _ in range(1000):
        took = eng.step_all()
        assert isinstance(took, bool)
        accepts += took
        eng.measure()
    assert eng.measure_step_counter == 1001 and eng.step_counter == 1001
    assert eng.real_params.shape == (1,) and eng.complex_params.shape == (0,)
    assert eng.real_mean.shape == (1,) and eng.covariance_matrix_real.shape == (1, 1)
    assert eng.covariance_matrix_complex is None
    assert eng.observables_names == ["abs_param_0", "param_0_squared"] and eng.observables_mean.shape == (2,)
    assert list(eng.energy) == ["total"] and abs(eng.energy["total"] - eng.real_params[0] ** 2) < 1e-12
    assert 300 < accepts < 560                       # the reference lands at 421-435 on its seeded runs
    assert accepts == eng.accept_stats()[0]
    # same regime as the reference's seeded anchors (tests/golden/seeded_readme_1real.npz): width ~0.52-0.57,
    # regularised "covariance" ~ Sigma + sigma^2 ~ 0.22-0.25
    assert 0.4 < eng.real_group_sampling_width < 0.7
    assert 0.15 < eng.covariance_matrix_real[0, 0] < 0.35
    assert eng.complex_group_sampling_width == 0.05


def test_pooled_shared_covariance_mode():
    """cov_mode='pooled': proposals shaped by one factor from the pooled covariance (dense 64-param config 4 regime,
    here on 4 real parameters)."""
    from metropolisengine_amd.distributed import moments_to_statistics, pooled_factor
    rng = np.random.default_rng(1)
    m = rng.standard_normal((4, 4))
    amat = m @ m.T / 4 + np.identity(4)
    eng = me.MetropolisEngine(me.DenseQuadratic(amat), None, [0.0] * 4, None, temp=1.0, n_chains=1 << 14, seed=21,
                              sampling_width=0.3, cov_mode="pooled")
    eng.step_all(600)
    st = moments_to_statistics(eng.pooled_moments(), 4, 0)
    eng.set_shared_factor(pooled_factor(st["covariance"], 4, 0))
    eng.step_all(600)
    st = moments_to_statistics(eng.pooled_moments(), 4, 0)
    want = 0.5 * np.linalg.inv(amat)                  # stationary covariance of exp(-x^T A x / T), T = 1
    assert np.all(np.abs(st["covariance"] - want) < 6 * np.max(np.abs(want)) * np.sqrt(2.0 / (1 << 14)))


def test_errors_on_gpu():
    with pytest.raises(NotImplementedError, match="identity proposal shape"):      # runtime-dimension kernels (beyond
        me.MetropolisEngine(me.IsoQuadratic(), initial_real_params=[0.0] * 100, initial_complex_params=[0j] * 30, temp=1.0,
                            cov_mode="pooled")      # build.MAX_COMPILED_DOF): no shared factor with complex parameters
    ref = me.MetropolisEngine(me.IsoQuadratic(), initial_real_params=[0.0] * 64, temp=1.0, n_chains=64)   # streamed per-chain
    assert ref.covariance_matrix_real.shape == (64, 64, 64)                                              # shapes (reference mode)
    eng = me.MetropolisEngine(me.IsoQuadratic(), initial_real_params=[0.0] * 64, temp=1.0, cov_mode="fixed",
                              n_chains=256)
    eng.step_all(3)
    eng.measure()
    with pytest.raises(NotImplementedError):
        eng.covariance_matrix_real
    with pytest.raises(ValueError):
        me.MetropolisEngine(me.IsoQuadratic(), initial_real_params=[0.0, 0.0], temp=1.0,
                            covariance_matrix_real=[[1.0, 2.0], [2.0, 1.0]])                     # not PSD (:270)


def test_full_size_properties_config2():
    """BASELINE config 2 at full size (2^20 chains x 16 real): determinism, acceptance, pooled variance."""
    n = 1 << 20
    kw = dict(temp=1.0, n_chains=n, seed=2026, sampling_width=0.3)
    a = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 16, None, **kw)
    b = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 16, None, **kw)
    a.step_all(400)
    b.step_all(200)
    b.step_all(200)
    ma, mb = a.pooled_moments(), b.pooled_moments()
    assert np.array_equal(ma, mb)   # the two-stage reduction has a fixed summation order: bitwise reproducible
    assert np.array_equal(a._get(0, 12345, 4096), b._get(0, 12345, 4096))
    from metropolisengine_amd.distributed import moments_to_statistics
    st = moments_to_statistics(ma, 16, 0)
    assert st["n_chains"] == n
    assert np.all(np.abs(np.diag(st["covariance"]) - 0.5) < 6 * 0.5 * np.sqrt(2.0 / n) + 2e-3)
    assert 0.2 < st["acceptance_rate"] < 0.45


def test_user_energy_plugin_matches_builtin_and_oracle():
    """BASELINE config 5 shape (2 real + 7 complex, hard wall |x0| >= 1): the user-callback plugin
    (examples/user_energy_cylinder.h), inlined and through a __device__ function pointer, against the built-in
    surrogate (bitwise, float32) and the oracle (float64, 1e-9)."""
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "user_energy_cylinder.h")
    coef = (1.0, 0.5, 1.0)
    real0, cplx0 = [0.1, 0.0], [0.05] * 7
    common = dict(temp=0.1, n_chains=2048, seed=2026)
    wall = me.AbsReal0AtLeast(1.0)
    builtin = me.MetropolisEngine(me.CylinderSurrogate(*coef), wall, real0, cplx0, **common)
    direct = me.MetropolisEngine(me.UserEnergy("cylinder", src, coef), wall, real0, cplx0, **common)
    indirect = me.MetropolisEngine(me.UserEnergy("cylinder", src, coef, indirect=True), wall, real0, cplx0, **common)
    # short horizon: the three code paths take the same decisions (float32 rounding may differ in the last bit)
    for eng in (builtin, direct, indirect):
        eng.step_all(3)
    for field in (0, 1, 2):
        ref = builtin._get(field)
        assert np.allclose(direct._get(field), ref, rtol=1e-5, atol=1e-6), field
        assert np.allclose(indirect._get(field), ref, rtol=1e-5, atol=1e-6), field
    # long horizon: float32 trajectories may diverge after a rounding-level accept flip, the statistics may not
    for eng in (builtin, direct, indirect):
        for _ in range(60):
            eng.step_all(10)
            eng.measure()
    from metropolisengine_amd.distributed import moments_to_statistics
    stats = [moments_to_statistics(eng.pooled_moments(), 2, 7) for eng in (builtin, direct, indirect)]
    for st in stats[1:]:
        assert abs(st["acceptance_rate"] - stats[0]["acceptance_rate"]) < 0.01
        assert np.allclose(np.diag(st["covariance"]), np.diag(stats[0]["covariance"]), rtol=0.15)
    assert np.all(np.abs(builtin._get(0)[:, 0]) < 1.0) and np.all(np.abs(indirect._get(0)[:, 0]) < 1.0)   # the wall
    # float64 plugin vs the oracle on the same Philox streams
    eng = me.MetropolisEngine(me.UserEnergy("cylinder", src, coef), wall, real0, cplx0, temp=0.1, n_chains=128,
                              seed=5, dtype="f64")
    ora = ManyChainOracle(2, 7, energies.cylinder_surrogate(2, 7, *coef), 128, seed=5, temp=0.1,
                          initial_real_params=real0, initial_complex_params=cplx0, reject=energies.wall_reject(1.0))
    for _ in range(55):
        eng.step_all(4)
        ora.step(4)
        eng.measure()
        ora.measure()
    assert np.allclose(eng._get(0), ora.x, rtol=0, atol=TOL)
    assert np.allclose(eng.covariance_matrix_complex, ora.cov_complex, rtol=0, atol=TOL)
    assert eng.accept_stats() == (ora.accepted, ora.proposed)


def test_dimensions_outside_the_prebuilt_set():
    """(3 real, 2 complex) is not in build.KERNEL_DIMS: the engine compiles / loads a kernel-set plugin on demand
    (prebuilt by __graft_entry__.build()) and must follow the oracle like any other size."""
    real0, cplx0 = [0.1, -0.2, 0.3], [0.1 + 0.1j, -0.2j]
    a, b = (1.0, 2.0, 0.5), (1.5, 3.0)
    eng = me.MetropolisEngine(me.DiagQuadratic(a, b), None, real0, cplx0, temp=0.8, n_chains=128, seed=77, dtype="f64")
    ora = ManyChainOracle(3, 2, energies.diag_quadratic(3, 2, a, b), 128, seed=77, temp=0.8,
                          initial_real_params=real0, initial_complex_params=cplx0)
    for _ in range(60):
        eng.step_all(3)
        ora.step(3)
        eng.measure()
        ora.measure()
    assert np.allclose(eng._get(0), ora.x, rtol=0, atol=TOL)
    assert np.allclose(eng.covariance_matrix_real, ora.cov_real, rtol=0, atol=TOL)
    assert np.allclose(eng.covariance_matrix_complex, ora.cov_complex, rtol=0, atol=TOL)
    assert eng.accept_stats() == (ora.accepted, ora.proposed)
