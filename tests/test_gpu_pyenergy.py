"""The reference's own way of passing an energy -- a Python callable, positional first (metropolis_engine.py:17, :20) -- on the
GPU: README.md:26-51, demo/toymodel_xypotentialwell.py and demo/toymodel_complex_and_real.py transcribed, the callables traced
into device functions (metropolisengine_amd/pyenergy.py), float64 against the many-chain oracle evaluating the SAME Python
functions chain by chain, on the same Philox streams."""
import numpy as np
import pytest

import metropolisengine_amd as me
from oracle.manychain import ManyChainOracle
from reference_style_energies import (LANDAU, WELL, energy_function, hundred_parameters, landau_dictionary, landau_total,
                                      narrow_wall, readme_energy, stiffer_well, wall, well_energy)

pytestmark = pytest.mark.gpu


def vectorised(fn, nr, nc):
    """The oracle's energy(x[n, D]) from the reference-style callable: one Python call per chain."""
    def energy(x):
        return np.array([complex(fn(row[:nr], row[nr:nr + nc] + 1j * row[nr + nc:])).real for row in x])
    return energy


def vectorised_reject(fn, nr, nc):
    return lambda x: np.array([bool(fn(row[:nr], row[nr:nr + nc] + 1j * row[nr + nc:])) for row in x])


def _follow(eng, ora, cycles, sweeps):
    for _ in range(cycles):
        eng.step_all(sweeps)
        ora.step(sweeps)
        eng.measure()
        ora.measure()
    assert np.allclose(eng._get(0), ora.x, rtol=0, atol=1e-9)
    assert np.allclose(eng.energy_total, ora.energy, rtol=0, atol=1e-9)
    assert np.allclose(eng._get(3), ora.mean, rtol=0, atol=1e-9)
    assert eng.accept_stats() == (ora.accepted, ora.proposed)


def test_readme_example_with_its_lambda():
    """README.md:26-51, the constructor call as written there."""
    n = 96
    eng = me.MetropolisEngine(lambda real_params, complex_params: energy_function(*real_params), initial_real_params=[0.0],
                              temp=.01, n_chains=n, seed=12345, dtype="f64")
    ora = ManyChainOracle(1, 0, vectorised(readme_energy, 1, 0), n, seed=12345, temp=.01, initial_real_params=[0.0])
    _follow(eng, ora, 60, 1)
    assert np.allclose(eng.covariance_matrix_real, ora.cov_real, rtol=0, atol=1e-9)       # past the 50-measure threshold
    assert eng.energy_term_names == ["total"]


def test_xy_well_demo_with_a_system_object():
    """demo/toymodel_xypotentialwell.py: the energy is a method of an object holding the constant."""
    n = 80
    eng = me.MetropolisEngine(lambda real_params, complex_params: WELL.calc_system_energy(real_params),
                              initial_real_params=np.array([0.0, 0.0]), temp=0.1, n_chains=n, seed=7, dtype="f64")
    ora = ManyChainOracle(2, 0, vectorised(well_energy, 2, 0), n, seed=7, temp=0.1, initial_real_params=[0.0, 0.0])
    _follow(eng, ora, 20, 10)


def test_landau_demo_with_its_energy_dictionary_and_a_python_wall():
    """demo/toymodel_complex_and_real.py:31-34: the dictionary of term callables, complex arithmetic with .conjugate(), group-wise
    stepping; plus a Python reject_condition given to the constructor (honoured here; dropped by the reference, quirk Q6)."""
    n = 72
    eng = me.MetropolisEngine(landau_dictionary(), wall, np.array([0.0, 0.0]), np.array([0 + 0j]), temp=0.1, n_chains=n,
                              seed=21, dtype="f64", sampling_width=0.3)
    assert eng.energy_term_names == ["area", "field"]
    ora = ManyChainOracle(2, 1, vectorised(landau_total, 2, 1), n, seed=21, temp=0.1, initial_real_params=[0.0, 0.0],
                          initial_complex_params=[0j], sampling_width=0.3, reject=vectorised_reject(wall, 2, 1))
    _follow(eng, ora, 12, 5)
    eng.step_real_group(3)
    ora.step(3, group="real")
    eng.step_complex_group(3)
    ora.step(3, group="complex")
    assert np.allclose(eng._get(0), ora.x, rtol=0, atol=1e-9)
    assert np.all(np.abs(eng._get(0)[:, 0]) < 1.0)
    ledger = eng.energy                                                # one row per term, named as in the dictionary
    x = eng._get(0)
    area = np.array([LANDAU.calc_area_energy(r[0], r[1]) for r in x])
    field = np.array([complex(LANDAU.calc_field_energy(r[0], r[1], r[2] + 1j * r[3])).real for r in x])
    assert np.allclose(ledger["area"], area, atol=1e-9) and np.allclose(ledger["field"], field, atol=1e-9)


def test_float32_and_the_one_launch_cycle_with_a_python_energy():
    n = 4096
    a = me.MetropolisEngine(lambda r, c: WELL.calc_system_energy(r), initial_real_params=[0.0, 0.0], temp=0.1, n_chains=n, seed=3)
    b = me.MetropolisEngine(lambda r, c: WELL.calc_system_energy(r), initial_real_params=[0.0, 0.0], temp=0.1, n_chains=n, seed=3)
    for _ in range(55):
        a.cycle(4)
        b.step_all(4)
        b.measure()
    assert a.fused_cycles() == 55
    for field in range(7):
        assert np.array_equal(a._get(field), b._get(field)), field
    var = a._get(0).var(axis=0)
    assert np.all(np.abs(var / 0.05 - 1.0) < 0.15)                      # Var x = T / (2 const) = 0.05


@pytest.mark.parametrize("cov_mode", ["fixed", "reference"])
def test_python_energy_with_100_parameters(cov_mode):
    """The reference has no limit on the number of parameters (metropolis_engine.py:41-60).  Built-in energies beyond 96 degrees
    of freedom run on the runtime-dimension kernels; a USER energy (here traced from Python) compiles the register-resident
    kernel set for its own size -- with cov_mode="reference" the per-chain 100 x 100 shapes stream (5 050 packed entries)."""
    n, seed = 70, 9
    x0 = list(np.linspace(-0.1, 0.1, 100))
    eng = me.MetropolisEngine(hundred_parameters, None, x0, None, temp=1.0, n_chains=n, seed=seed, dtype="f64",
                              sampling_width=0.05, cov_mode=cov_mode)
    ora = ManyChainOracle(100, 0, vectorised(hundred_parameters, 100, 0), n, seed=seed, temp=1.0, initial_real_params=x0,
                          sampling_width=0.05, adapt_shape=cov_mode == "reference")
    cycles = 54 if cov_mode == "reference" else 8
    for _ in range(cycles):
        eng.step_all(2)
        ora.step(2)
        eng.measure()
        ora.measure()
    for sweeps in (1, 3):
        eng.step_all(sweeps)
        ora.step(sweeps)
    assert np.allclose(eng._get(0), ora.x, rtol=0, atol=1e-8)
    assert np.allclose(eng.energy_total, ora.energy, rtol=0, atol=1e-8)
    assert eng.accept_stats() == (ora.accepted, ora.proposed)
    if cov_mode == "reference":
        fr, _ = eng.proposal_factors()
        assert np.allclose(fr, ora.factor_real, rtol=0, atol=1e-8)


def test_set_energy_function_and_set_reject_condition():
    """The reference's setters (metropolis_engine.py:134-146) on a live engine: another Python energy on the same parameter
    space (the ledger is re-evaluated at the current state), a Python wall installed after construction (the reference's only
    working way, quirk Q6), a term dictionary replacing a single function (one ledger row becomes two), and built-in kinds."""
    n = 64
    eng = me.MetropolisEngine(readme_energy, initial_real_params=[0.0], temp=.01, n_chains=n, seed=5, dtype="f64")
    ora = ManyChainOracle(1, 0, vectorised(readme_energy, 1, 0), n, seed=5, temp=.01, initial_real_params=[0.0])
    _follow(eng, ora, 5, 3)
    eng.set_energy_function(stiffer_well)
    ora.energy_fn = vectorised(stiffer_well, 1, 0)
    ora.energy = ora.energy_fn(ora.x)
    assert np.allclose(eng.energy_total, 2.0 * eng._get(0)[:, 0] ** 2, atol=1e-14)          # re-evaluated, not stale
    _follow(eng, ora, 5, 3)
    eng.set_energy_function(readme_energy)
    ora.energy_fn = vectorised(readme_energy, 1, 0)
    ora.energy = ora.energy_fn(ora.x)
    eng.set_reject_condition(narrow_wall)
    ora.reject_fn = vectorised_reject(narrow_wall, 1, 0)
    _follow(eng, ora, 20, 3)
    assert np.all(np.abs(eng._get(0)[:, 0]) < 0.05) and np.any(np.abs(eng._get(0)[:, 0]) > 0.03)
    eng.set_reject_condition(None)
    ora.reject_fn = None
    _follow(eng, ora, 5, 3)

    # a single function -> the demo's term dictionary: the ledger grows from one row to two, the names follow
    mixed = me.MetropolisEngine(landau_total, None, [0.2, 0.3], [0.1 + 0.1j], temp=0.1, n_chains=n, seed=6, dtype="f64")
    mixed.step_all(5)
    assert mixed.energy_term_names == ["total"]
    total = mixed.energy_total
    mixed.set_energy_function(landau_dictionary())
    assert mixed.energy_term_names == ["area", "field"] and set(mixed.energy) == {"area", "field"}
    assert np.allclose(mixed.energy["area"] + mixed.energy["field"], total, atol=1e-12)
    mixed.step_real_group(2)
    mixed.step_all(2)
    mixed.measure()

    # built-in kinds through the same C entry point
    built = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.1, 0.2], None, temp=1.0, n_chains=n, seed=7, dtype="f64")
    built.step_all(3)
    built.set_energy_function(me.DiagQuadratic((1.0, 3.0)))
    x = built._get(0)
    assert np.allclose(built.energy_total, x[:, 0] ** 2 + 3.0 * x[:, 1] ** 2, atol=1e-14)
    with pytest.raises(TypeError):
        built.set_reject_condition(narrow_wall)               # a Python predicate needs a Python energy
