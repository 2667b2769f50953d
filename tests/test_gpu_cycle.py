"""me_cycle / engine.cycle(k): k sweeps + measure in ONE launch (k_cycle) must give exactly what step_all(k); measure()
gives -- the reference's driver loop, README.md:41-44 -- on BASELINE configs 3 and 5, before and after the 50-measure
threshold that switches the per-chain adaptive covariance on (metropolis_engine.py:389).  float32: bitwise, every field.
float64: against the numpy oracle at 1e-9 (and bitwise against the two-launch form)."""
import os

import numpy as np
import pytest

import metropolisengine_amd as me
from metropolisengine_amd import _capi
from oracle import energies
from oracle.manychain import ManyChainOracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = (_capi.FIELD_PARAMS, _capi.FIELD_ENERGY, _capi.FIELD_WIDTH, _capi.FIELD_MEAN, _capi.FIELD_COV,
          _capi.FIELD_OBS_MEAN, _capi.FIELD_FACTOR)


def _config3(dtype, n, **kw):
    a = b = (1.0, 2.0, 4.0, 8.0)
    return me.MetropolisEngine(me.DiagQuadratic(a, b), None, [0.0] * 4, [0j] * 4, temp=1.0, n_chains=n, seed=2026, dtype=dtype, **kw)


def _config5(dtype, n, **kw):
    src = os.path.join(ROOT, "examples", "user_energy_cylinder.h")
    return me.MetropolisEngine(me.UserEnergy("cylinder", src, (1.0, 0.5, 1.0)), me.AbsReal0AtLeast(1.0), [0.1, 0.0],
                               [0.05] * 7, temp=0.1, n_chains=n, seed=2026, dtype=dtype, **kw)


@pytest.mark.parametrize("make", [_config3, _config5], ids=["config3_4real_4complex", "config5_cylinder_plugin"])
@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_cycle_equals_step_then_measure_bitwise(make, dtype):
    n, k, cycles = 4096 + 37, 10, 58          # ragged last tile; 58 measures: the per-chain factors go live at 51
    fused, split = make(dtype, n), make(dtype, n)
    for i in range(cycles):
        fused.cycle(k)
        split.step_all(k)
        split.measure()
        if i in (0, 49, 50, 51, cycles - 1):
            for field in FIELDS:
                assert np.array_equal(fused._get(field), split._get(field)), "field %d differs after cycle %d" % (field, i)
    assert fused.fused_cycles() == cycles and split.fused_cycles() == 0
    assert fused.accept_stats() == split.accept_stats()
    assert fused.measure_step_counter == split.measure_step_counter == cycles + 1
    assert fused.step_counter == split.step_counter
    # mixing the two call styles on one engine continues the same trajectory
    fused.step_all(3)
    fused.measure()
    split.cycle(3)
    for field in FIELDS:
        assert np.array_equal(fused._get(field), split._get(field))


def test_cycle_follows_the_oracle_float64():
    a = b = (1.0, 2.0, 4.0, 8.0)
    n = 192
    eng = _config3("f64", n)
    ora = ManyChainOracle(4, 4, energies.diag_quadratic(4, 4, a, b), n, seed=2026, temp=1.0, initial_real_params=[0.0] * 4,
                          initial_complex_params=[0j] * 4)
    for _ in range(56):
        eng.cycle(5)
        ora.step(5)
        ora.measure()
    assert eng.fused_cycles() == 56
    assert np.allclose(eng._get(_capi.FIELD_PARAMS), ora.x, rtol=0, atol=1e-9)
    assert np.allclose(eng._get(_capi.FIELD_MEAN), ora.mean, rtol=0, atol=1e-9)
    assert np.allclose(eng.covariance_matrix_real, ora.cov_real, rtol=0, atol=1e-9)
    assert np.allclose(eng.covariance_matrix_complex, ora.cov_complex, rtol=0, atol=1e-9)
    assert np.allclose(eng.observables_mean, ora.observables_mean, rtol=0, atol=1e-9)
    assert eng.accept_stats() == (ora.accepted, ora.proposed)


def test_cycle_falls_back_to_two_launches_where_no_fused_kernel_exists():
    """Engines without a register-resident per-chain covariance (64 real parameters: matrix-core step kernel) and engines
    with a shared factor run me_cycle as a step launch + a measure launch: same results, fused_cycles() stays 0."""
    m = np.random.default_rng(5).standard_normal((64, 64))
    amat = m @ m.T / 64 + np.identity(64)
    e1 = me.MetropolisEngine(me.DenseQuadratic(amat), None, [0.0] * 64, None, temp=1.0, n_chains=512, seed=3, cov_mode="fixed")
    e2 = me.MetropolisEngine(me.DenseQuadratic(amat), None, [0.0] * 64, None, temp=1.0, n_chains=512, seed=3, cov_mode="fixed")
    for _ in range(3):
        e1.cycle(4)
        e2.step_all(4)
        e2.measure()
    assert e1.fused_cycles() == 0
    assert np.array_equal(e1._get(_capi.FIELD_PARAMS), e2._get(_capi.FIELD_PARAMS))
    assert np.array_equal(e1._get(_capi.FIELD_MEAN), e2._get(_capi.FIELD_MEAN))


def test_cycle_records_the_time_series_like_measure():
    eng = _config3("f64", 64, trace_chains=2)
    ref = _config3("f64", 64, trace_chains=2)
    for _ in range(5):
        eng.cycle(3)
        ref.step_all(3)
        ref.measure()
    assert np.array_equal(eng.trace(), ref.trace()) and eng.trace().shape[0] == 5


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_cycle_with_energy_dictionaries_two_ledger_mode_and_wall(dtype):
    """The fused cycle keeps the energy ledger exactly as the step kernel does: the Landau toy as a term dictionary (two
    ledger rows, metropolis_engine.py:111-116), the reference's two-ledger mode (quirk Q5: step_all decides against
    energy_total only, :252-255) and a hard wall (:247-249) -- bitwise against step_all(k); measure(), every field."""
    n, k = 1000, 4

    def landau_terms(**kw):
        return me.MetropolisEngine(me.LandauToy(1.0, -1.0, 0.5, terms=True), None, [0.5, 0.5], [0.3 + 0.1j], temp=0.1,
                                   n_chains=n, seed=5, dtype=dtype, **kw)

    def walled(**kw):
        return me.MetropolisEngine(me.DiagQuadratic((0.2, 1.0), (1.0, 2.0)), me.AbsReal0AtLeast(0.4), [0.1, 0.0], [0.1j, 0.2],
                                   temp=1.0, n_chains=n, seed=6, dtype=dtype, sampling_width=0.3, **kw)

    for make, kw, fields in ((landau_terms, {}, FIELDS), (walled, {}, FIELDS),
                             (walled, {"reference_energy_ledgers": True}, FIELDS + (_capi.FIELD_ENERGY_TOTAL,))):
        fused, split = make(**kw), make(**kw)
        if kw:                                   # make the two ledgers differ: group steps touch energy[term] only
            for e in (fused, split):
                e.step_real_group(3)
                e.step_complex_group(2)
        for _ in range(54):
            fused.cycle(k)
            split.step_all(k)
            split.measure()
        assert fused.fused_cycles() == 54
        for field in fields:
            assert np.array_equal(fused._get(field), split._get(field)), (make.__name__, kw, field)
        assert fused.accept_stats() == split.accept_stats()
        if make is walled:
            assert np.all(np.abs(fused._get(0)[:, 0]) < 0.4)
