"""GPU tests of energy dictionaries (metropolis_engine.py:111-116, :152-162, :209-239): the energy ledger keeps one row
per term and group steps re-evaluate only their group's terms.  Golden trajectories recorded from the reference with the
demo's Landau term dictionary are replayed in float64; a term-wise user plugin is checked against the built-in."""
import os

import numpy as np
import pytest

import metropolisengine_amd as me
from metropolisengine_amd import _capi
from oracle import scenarios

pytestmark = pytest.mark.gpu
TOL = 1e-9
KIND = {"all": _capi.STEP_ALL, "real": _capi.STEP_REAL_GROUP, "complex": _capi.STEP_COMPLEX_GROUP}
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_golden_term_ledger_on_gpu(golden_dir):
    """groups_landau_terms: step_real_group / step_complex_group with the term dictionary; the reference's self.energy
    after every step is the ledger the kernels must hold."""
    spec = scenarios.SCENARIOS["groups_landau_terms"]
    gold = np.load(os.path.join(golden_dir, "traj_groups_landau_terms.npz"))
    names = [str(n) for n in gold["term_names"]]
    assert sorted(names) == ["area", "field"]
    eng = me.MetropolisEngine(me.LandauToy(1.0, -1.0, 0.5, terms=True), None, spec["real"], spec["cplx"],
                              temp=spec["temp"], n_chains=1, dtype="f64")
    assert eng.energy_term_names == ["field", "area"]
    t = 0
    for k in range(spec["n_measures"]):
        for op in scenarios.ops(spec)[:-1]:
            eng.step_injected(gold["normals"][t:t + 1, None, :], gold["uniforms"][t:t + 1, :1], kind=KIND[op])
            ledger = eng.energy
            for i, name in enumerate(names):
                assert abs(ledger[name] - gold["energy_terms"][t, i]) < TOL, (t, op, name)
            assert np.allclose(eng.real_params, gold["real_params"][t], rtol=0, atol=TOL), (t, op)
            assert np.allclose(eng.complex_params, gold["complex_params"][t], rtol=0, atol=TOL), (t, op)
            assert abs(eng.real_group_sampling_width - gold["real_width"][t]) < TOL
            assert abs(eng.complex_group_sampling_width - gold["complex_width"][t]) < TOL
            t += 1
        eng.measure()
    # the DataFrame carries one "<term>_energy" column per term (:468-469)
    frame = eng.time_series_frame()
    cols = [str(c) for c in gold["df_columns"]]
    assert sorted(frame.columns) == sorted(cols)
    for name in ("area_energy", "field_energy", "param_0", "real_group_sampling_width", "abs_param_2"):
        want = gold["df_values"][:, cols.index(name)]
        assert np.allclose(np.asarray(frame[name], dtype=np.complex128), want, rtol=0, atol=TOL), name
    series = eng.energy_time_series
    assert set(series) == {"field", "area"} and len(series["area"]) == spec["n_measures"]


def test_reference_golden_step_all_with_terms(golden_dir):
    """landau_toy: the same dictionary driven through step_all; the total is the sum of the ledger (:158-162)."""
    spec = scenarios.SCENARIOS["landau_toy"]
    gold = np.load(os.path.join(golden_dir, "traj_landau_toy.npz"))
    eng = me.MetropolisEngine(me.LandauToy(1.0, -1.0, 0.5, terms=True), None, spec["real"], spec["cplx"],
                              temp=spec["temp"], n_chains=1, dtype="f64")
    t = 0
    for k in range(spec["n_measures"]):
        for _ in range(spec["steps_per_measure"]):
            eng.step_injected(gold["normals"][t:t + 1, None, :], gold["uniforms"][t:t + 1, :1])
            assert abs(eng.energy_total - gold["energy_total"][t]) < TOL, t
            assert np.allclose(eng.real_params, gold["real_params"][t], rtol=0, atol=TOL), t
            t += 1
        eng.measure()
        assert np.allclose(eng.covariance_matrix_real, gold["cov_real"][k], rtol=0, atol=TOL)


def _drive(eng, cycles):
    for _ in range(cycles):
        eng.step_real_group()
        eng.step_complex_group()
        eng.step_all(2)
        eng.step_real_group(3)
        eng.measure()


def test_term_plugin_matches_builtin_and_ledger_stays_consistent():
    n = 4096 + 5
    kw = dict(temp=0.1, n_chains=n, seed=17, dtype="f64")
    src = os.path.join(REPO, "examples", "user_energy_landau_terms.h")
    user = me.UserEnergy("landau_terms", src, (1.0, -1.0, 0.5), term_names=("field", "area"))
    a = me.MetropolisEngine(me.LandauToy(1.0, -1.0, 0.5, terms=True), None, [0.0, 0.0], [0j], **kw)
    b = me.MetropolisEngine(user, None, [0.0, 0.0], [0j], **kw)
    assert b.energy_term_names == ["field", "area"]
    _drive(a, 30)
    _drive(b, 30)
    assert np.allclose(a._get(0), b._get(0), rtol=0, atol=1e-12)
    for name in ("field", "area"):
        assert np.allclose(a.energy[name], b.energy[name], rtol=0, atol=1e-12)
    assert a.accept_stats() == b.accept_stats()
    # the cached terms equal a fresh evaluation at the current state
    x = a._get(0)
    a2 = x[:, 2] ** 2 + x[:, 3] ** 2
    assert np.allclose(a.energy["field"], x[:, 0] * x[:, 1] * (-a2 + 0.5 * a2 * a2), rtol=0, atol=1e-12)
    assert np.allclose(a.energy["area"], (1 - x[:, 0]) ** 2 + (1 - x[:, 1]) ** 2, rtol=0, atol=1e-12)
    cached = a._get(_capi.FIELD_ENERGY).copy()
    a.initialize_energy_dict()
    assert np.allclose(a._get(_capi.FIELD_ENERGY), cached, rtol=0, atol=1e-12)


def test_terms_and_total_forms_sample_the_same_distribution_f32():
    n = 1 << 14
    kw = dict(temp=0.1, n_chains=n, seed=23)
    terms = me.MetropolisEngine(me.LandauToy(1.0, -1.0, 0.5, terms=True), None, [0.0, 0.0], [0j], **kw)
    total = me.MetropolisEngine(me.LandauToy(1.0, -1.0, 0.5), None, [0.0, 0.0], [0j], **kw)
    for eng in (terms, total):
        _drive(eng, 150)
    xa, xb = terms._get(0), total._get(0)
    se = np.sqrt((xa.var(axis=0) + xb.var(axis=0)) / n)
    assert np.all(np.abs(xa.mean(axis=0) - xb.mean(axis=0)) < 6 * se + 1e-6)
    assert abs(terms.acceptance_rate() - total.acceptance_rate()) < 0.01
    assert np.allclose(terms.energy_total, terms.energy["field"] + terms.energy["area"], rtol=1e-6)


def test_term_names_must_match_the_device_ledger():
    src = os.path.join(REPO, "examples", "user_energy_landau_terms.h")
    with pytest.raises(ValueError):
        me.MetropolisEngine(me.UserEnergy("landau_terms", src, (1.0, -1.0, 0.5)), None, [0.0, 0.0], [0j], temp=0.1,
                            n_chains=64)


def test_magnitude_phase_and_checkpoint_with_terms():
    """The magnitude-phase sampler compares the complex group's terms only (:183-189); in float64 its trajectory equals
    the single-function engine's (the untouched "area" term cancels).  state_dict carries every ledger row."""
    kw = dict(temp=0.1, n_chains=512, seed=5, dtype="f64", complex_sample_method="magnitude-phase")
    terms = me.MetropolisEngine(me.LandauToy(1.0, -1.0, 0.5, terms=True), None, [0.3, 0.2], [0.4 + 0.1j], **kw)
    total = me.MetropolisEngine(me.LandauToy(1.0, -1.0, 0.5), None, [0.3, 0.2], [0.4 + 0.1j], **kw)
    for eng in (terms, total):
        for _ in range(60):                    # > 50 measures: the covariance feeds the magnitude step
            eng.step_real_group()
            eng.step_complex_group()
            eng.measure()
    same = np.all(np.abs(terms._get(0) - total._get(0)) < 1e-9, axis=1)
    assert same.mean() > 0.99                  # a flipped near-tie decision would show up as a diverged chain
    assert np.allclose(terms.energy_total[same], total.energy_total[same], rtol=0, atol=1e-9)
    state = terms.state_dict()
    assert state["energy"].shape == (512, 2)
    clone = me.MetropolisEngine(me.LandauToy(1.0, -1.0, 0.5, terms=True), None, [0.0, 0.0], [0j], **kw)
    clone.load_state_dict(state)
    for eng in (terms, clone):
        eng.step_complex_group(3)
        eng.step_all(2)
    assert np.array_equal(terms._get(0), clone._get(0))
    assert np.array_equal(terms._get(_capi.FIELD_ENERGY), clone._get(_capi.FIELD_ENERGY))
