"""GPU tests of the constructor surface the reference documents (metropolis_engine.py:17-29): warm-start covariance
matrices, target acceptance, default temperature, names, field writes, and the numeric-failure channel."""
import numpy as np
import pytest

import metropolisengine_amd as me
from metropolisengine_amd import _capi
from oracle import energies
from oracle.manychain import ManyChainOracle
from oracle.reference_chain import adaptation_constants

pytestmark = pytest.mark.gpu
TOL = 1e-9


def test_warm_start_covariances_shape_the_first_proposals():
    """covariance_matrix_real / covariance_matrix_complex given to the constructor (:26-27, :63-70): one shared
    Cholesky factor (real block, and conj(K) for the complex block, quirk Q3) until the 51st measure."""
    nr, nc = 2, 2
    c_real = np.array([[2.0, 0.6], [0.6, 1.0]])
    c_cplx = np.array([[1.5, 0.3 - 0.4j], [0.3 + 0.4j, 0.8]])
    real0, cplx0 = [0.1, -0.2], [0.2 + 0.1j, -0.1j]
    a, b = (1.0, 2.0), (1.5, 3.0)
    eng = me.MetropolisEngine(me.DiagQuadratic(a, b), None, real0, cplx0, covariance_matrix_real=c_real,
                              covariance_matrix_complex=c_cplx, temp=1.0, n_chains=160, seed=31, dtype="f64")
    ora = ManyChainOracle(nr, nc, energies.diag_quadratic(nr, nc, a, b), 160, seed=31, temp=1.0,
                          initial_real_params=real0, initial_complex_params=cplx0, covariance_matrix_real=c_real,
                          covariance_matrix_complex=c_cplx)
    assert np.allclose(eng.covariance_matrix_real[0], c_real) and np.allclose(eng.covariance_matrix_complex[0], c_cplx)
    fr, fc = eng.proposal_factors()
    assert np.allclose(fr[0] @ fr[0].T, c_real) and np.allclose(fc[0] @ fc[0].conj().T, np.conj(c_cplx))
    for k in range(70):                       # crosses the switch from the shared to the per-chain factors
        eng.step_all(3)
        ora.step(3)
        eng.measure()
        ora.measure()
        if k in (10, 49, 50, 69):
            assert np.allclose(eng._get(0), ora.x, rtol=0, atol=TOL), k
    assert np.allclose(eng.covariance_matrix_complex, ora.cov_complex, rtol=0, atol=TOL)
    assert eng.accept_stats() == (ora.accepted, ora.proposed)


def test_target_acceptance_and_constants():
    for target in (0.2, 0.5):
        eng = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 4, None, target_acceptance=target, temp=1.0,
                                  n_chains=1 << 14, seed=3, sampling_width=0.3)
        alpha, m, ratio = adaptation_constants(4, 0, target)
        assert abs(eng.alpha - alpha) < 1e-12 and eng.m == m and abs(eng.ratio - ratio) < 1e-12
        assert eng.target_acceptance == target
        eng.step_all(3000)
        before = eng.accept_stats()
        eng.step_all(300)
        after = eng.accept_stats()
        rate = (after[0] - before[0]) / (after[1] - before[1])
        assert abs(rate - target) < 0.05      # Robbins-Monro drives every chain's acceptance to the target


def test_default_temperature_is_zero():
    """temp defaults to 0 (:17): uphill moves are never accepted (:331-332), the energy can only decrease."""
    eng = me.MetropolisEngine(me.DiagQuadratic((1.0, 3.0)), initial_real_params=[1.0, -1.0], n_chains=2048, seed=1)
    assert eng.temp == 0
    e0 = eng.energy_total.copy()
    eng.step_all(50)
    e1 = eng.energy_total
    assert np.all(e1 <= e0 + 1e-6) and np.mean(e1 < e0) > 0.5
    eng.step_all(50)
    assert np.all(eng.energy_total <= e1 + 1e-6)


def test_names_and_width_list():
    eng = me.MetropolisEngine(me.DiagQuadratic((1.0,), (2.0,)), None, [0.0], [0j], params_names=["amplitude", "field"],
                              temp=1.0, n_chains=4)
    assert eng.params_names == ["amplitude", "field"]
    assert eng.observables_names == ["abs_param_0", "abs_param_1", "param_0_squared"]
    assert eng.num_real_params == 1 and eng.num_complex_params == 1 and eng.param_space_dims == 2
    pure = me.MetropolisEngine(me.IsoQuadratic(1.0), None, None, [0j, 0j], sampling_width=[0.1, 0.2], temp=1.0, n_chains=4)
    assert np.allclose(pure.complex_group_sampling_width, 0.2) and pure.real_group_sampling_width == 0.1   # :94-95
    with pytest.raises(ValueError):           # quirk Q7: the list form breaks mixed engines in the reference
        me.MetropolisEngine(me.DiagQuadratic((1.0,), (2.0,)), None, [0.0], [0j], sampling_width=[0.1, 0.2], temp=1.0)
    eng.set_initial_sampling_width(0.7)       # the reference stores it in an attribute nothing reads (:148-149)
    assert eng.group_sampling_width == 0.7 and np.allclose(eng.sampling_width, 0.05)


def test_field_writes_and_energy_recompute():
    n = 300
    eng = me.MetropolisEngine(me.DiagQuadratic((1.0, 2.0)), None, [0.0, 0.0], None, temp=1.0, n_chains=n, seed=2)
    x = np.random.default_rng(0).standard_normal((n, 2))
    eng._set(0, x)
    eng._check(eng._lib.me_recompute_energy(eng._handle))
    assert np.allclose(eng._get(0), x.astype(np.float32), atol=1e-7)
    assert np.allclose(eng.energy_total, x[:, 0] ** 2 + 2 * x[:, 1] ** 2, rtol=1e-5)
    part = eng._get(0, 100, 50)               # sub-range reads
    assert np.array_equal(part, eng._get(0)[100:150])
    with pytest.raises(ValueError):
        eng._get(0, 290, 20)


def test_numeric_failures_surface_as_exceptions():
    """Per-chain failure flags are the analogue of the reference's exceptions: a non-finite
    energy (here: the cylinder surrogate's pole at |x0| = 1 without the hard wall) raises at the next sync."""
    eng = me.MetropolisEngine(me.CylinderSurrogate(1.0, 0.5, 1.0), None, [0.999, 0.0], [0.05] * 7, temp=5.0,
                              n_chains=4096, seed=1, sampling_width=0.001)
    x = eng._get(0)
    x[:16, 0] = 1.0                            # 1 - x0^2 = 0 -> energy inf for the chains sitting on the pole
    eng._set(0, x)
    with pytest.raises(FloatingPointError):
        eng._check(eng._lib.me_recompute_energy(eng._handle))
        eng.sync()
    eng.sync()                                 # the flags are cleared once reported


def test_set_reject_condition_after_construction():
    """The reference's constructor drops ``reject_condition`` (quirk Q6); its users install the wall with
    ``set_reject_condition`` (metropolis_engine.py:142-146).  Both routes must give the same chain."""
    args = (me.DiagQuadratic((0.5,), (1.0,)), )
    kw = dict(initial_real_params=[0.0], initial_complex_params=[0.1j], temp=1.0, n_chains=512, seed=4)
    ctor = me.MetropolisEngine(*args, reject_condition=me.AbsReal0AtLeast(0.25), **kw)
    setter = me.MetropolisEngine(*args, **kw)
    setter.set_reject_condition(me.AbsReal0AtLeast(0.25))
    free = me.MetropolisEngine(*args, **kw)
    for eng in (ctor, setter, free):
        eng.step_all(200)
    assert np.array_equal(ctor._get(0), setter._get(0)) and ctor.accept_stats() == setter.accept_stats()
    assert np.all(np.abs(setter._get(0)[:, 0]) < 0.25) and np.any(np.abs(free._get(0)[:, 0]) >= 0.25)
    setter.set_reject_condition(None)                 # lifting the wall
    setter.step_all(400)
    assert np.any(np.abs(setter._get(0)[:, 0]) >= 0.25)
    with pytest.raises(TypeError):
        setter.set_reject_condition(lambda r, c: False)
    with pytest.raises(ValueError):
        me.MetropolisEngine(me.IsoQuadratic(1.0), me.AbsReal0AtLeast(1.0), None, [0j], temp=1.0, n_chains=8)


def test_largest_register_resident_dimension():
    """96 real degrees of freedom is the largest kernel set built by default (build.MAX_REGISTER_DOF).  The sampler must
    still be right there: stationary variance T / (2 a) per coordinate."""
    n, a = 1 << 12, 2.0
    eng = me.MetropolisEngine(me.IsoQuadratic(a), initial_real_params=[0.0] * 96, temp=1.0, n_chains=n, seed=31,
                              sampling_width=0.08, cov_mode="fixed")
    eng.step_all(1500)
    x = eng._get(0)
    assert x.shape == (n, 96)
    var = x.var(axis=0)
    assert np.all(np.abs(var - 0.25) < 6 * 0.25 * np.sqrt(2.0 / n))
    assert np.all(np.abs(x.mean(axis=0)) < 6 * np.sqrt(0.25 / n))
    assert 0.15 < eng.acceptance_rate() < 0.5
    eng.measure()
    assert np.allclose(eng.real_mean[0], (x[0] + 0.0) / 2, atol=1e-6)      # mean of the initial point and the state
    # beyond build.MAX_COMPILED_DOF the default cov_mode (per-chain shapes) runs on the runtime-dimension set
    big = me.MetropolisEngine(me.IsoQuadratic(a), initial_real_params=[0.0] * 200, temp=1.0, n_chains=64)
    big.step_all(2)
    big.measure()
    assert big.cov_mode == "reference" and big.covariance_matrix_real.shape == (64, 200, 200)
    with pytest.raises(RuntimeError):               # ... what it does not have there is one shared factor for spaces with complex parameters
        me.MetropolisEngine(me.IsoQuadratic(a), initial_real_params=[0.0] * 100, initial_complex_params=[0j] * 50, temp=1.0, n_chains=64,
                            cov_mode="pooled")


def test_split_pooled_moments_report_the_state_at_begin():
    """me_pooled_moments_begin/_end: the reduction is enqueued behind the work already queued, later steps overlap the
    copy, and the result is the moment vector of the state AT begin -- bitwise the blocking call's."""
    eng = me.MetropolisEngine(me.DiagQuadratic((1.0, 2.0), (3.0,)), None, [0.2, -0.1], [0.1 + 0.3j], temp=1.0,
                              n_chains=(1 << 15) + 17, seed=3)
    eng.step_all(25)
    blocking = eng.pooled_moments()
    eng.pooled_moments_begin()
    with pytest.raises(RuntimeError):
        eng.pooled_moments_begin()                 # one in flight per engine
    with pytest.raises(RuntimeError):
        eng.pooled_moments()
    eng.step_all(40)                               # runs while the result travels; does not disturb it
    eng.measure()
    split = eng.pooled_moments_end()
    assert np.array_equal(split, blocking)
    with pytest.raises(RuntimeError):
        eng.pooled_moments_end()
    later = eng.pooled_moments()
    assert later[-1] == blocking[-1] + 40 * eng.n_chains and not np.array_equal(later[1:5], blocking[1:5])
    from metropolisengine_amd.distributed import pooled_statistics_begin, pooled_statistics_end, moments_to_statistics
    pooled_statistics_begin(eng)
    eng.step_all(3)
    stats = pooled_statistics_end(eng)
    want = moments_to_statistics(later, 2, 1)
    assert np.array_equal(stats["covariance"], want["covariance"])


def test_partial_get_set_of_the_tile_major_packed_fields():
    """ME_FIELD_COV / ME_FIELD_FACTOR are tile-major on the device ([64-chain tile][entry][lane]); me_get / me_set of any
    chain range -- inside one tile, across tile borders, up to the ragged last tile -- round-trips and leaves every
    other chain untouched."""
    n = 64 * 3 + 21
    a = (1.0, 2.0)
    eng = me.MetropolisEngine(me.DiagQuadratic(a, a), None, [0.0] * 2, [0j] * 2, temp=1.0, n_chains=n, seed=5, dtype="f64")
    for _ in range(55):
        eng.step_all(3)
        eng.measure()
    for field in (_capi.FIELD_COV, _capi.FIELD_FACTOR):
        before = eng._get(field)
        assert before.shape == (n, 2 * 3 // 2 + 4)
        for begin, count in ((5, 10), (60, 10), (0, 64), (100, 113), (n - 21, 21), (n - 1, 1)):
            assert np.array_equal(eng._get(field, begin, count), before[begin:begin + count])
            new = before[begin:begin + count] + 1.0 + np.arange(count)[:, None]
            eng._set(field, new, chain_begin=begin)
            after = eng._get(field)
            assert np.array_equal(after[begin:begin + count], new)
            mask = np.ones(n, dtype=bool)
            mask[begin:begin + count] = False
            assert np.array_equal(after[mask], before[mask])
            eng._set(field, before[begin:begin + count], chain_begin=begin)
        assert np.array_equal(eng._get(field), before)
