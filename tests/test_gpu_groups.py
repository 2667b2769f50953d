"""GPU tests of the step kinds beyond step_all: group-wise stepping of mixed engines (me_step_kind REAL/COMPLEX
group, metropolis_engine.py:209-239 called directly) and the magnitude-phase complex sampler (:168-207).

float64 engines follow the many-chain oracle on identical Philox streams (1e-9) and replay the reference's own golden
trajectories through me_step_injected; float32 engines are checked statistically.

A driver that MIXES step_all() with group steps on one mixed engine makes the reference compare against stale energies
(step_all keeps ``energy_total``, group steps keep ``energy[term]``: SURVEY.md quirk Q5).  By default the GPU engine
keeps one coherent energy per chain; with ``reference_energy_ledgers=True`` (ME_FLAG_REFERENCE_ENERGY_LEDGERS) it keeps
the reference's two ledgers, and the golden scenario ``groups_2real_2complex`` -- which mixes both call styles -- is
replayed through the HIP kernels like the others (test_reference_golden_mixed_call_styles_stale_ledgers).
"""
import os

import numpy as np
import pytest

import metropolisengine_amd as me
from metropolisengine_amd import _capi
from metropolisengine_amd.distributed import moments_to_statistics
from oracle import energies, scenarios
from oracle.manychain import ManyChainOracle

pytestmark = pytest.mark.gpu
TOL = 1e-9

CASES = {
    "groups_mixed": dict(nr=2, nc=2, method="multivariate-gaussian", ops=("real", "complex", "real", "measure"),
                         spec=me.DiagQuadratic((1.0, 2.0), (1.5, 3.0)),
                         energy=energies.diag_quadratic(2, 2, (1.0, 2.0), (1.5, 3.0)), temp=1.0,
                         real=[0.1, -0.1], cplx=[0.2 + 0.1j, -0.1 + 0.3j], cycles=70),
    "all_then_groups": dict(nr=2, nc=2, method="multivariate-gaussian", ops=("all", "all", "real", "complex", "all", "measure"),
                            spec=me.DiagQuadratic((1.0, 2.0), (1.5, 3.0)),
                            energy=energies.diag_quadratic(2, 2, (1.0, 2.0), (1.5, 3.0)), temp=1.0,
                            real=[0.1, -0.1], cplx=[0.2 + 0.1j, -0.1 + 0.3j], cycles=60),
    "magphase_mixed": dict(nr=1, nc=1, method="magnitude-phase", ops=("real", "complex", "measure"),
                           spec=me.DiagQuadratic((0.5,), (1.0,)), energy=energies.diag_quadratic(1, 1, (0.5,), (1.0,)),
                           temp=0.5, real=[0.2], cplx=[0.3 + 0.1j], cycles=80),
    "magphase_pure_complex": dict(nr=0, nc=3, method="magnitude-phase", ops=("complex", "all", "complex", "measure"),
                                  spec=me.DiagQuadratic((), (1.0, 2.0, 0.5)),
                                  energy=energies.diag_quadratic(0, 3, (), (1.0, 2.0, 0.5)), temp=0.5,
                                  real=None, cplx=[0.3 + 0.1j, 0.2j, 0.0], cycles=70),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_f64_step_kinds_follow_the_oracle(name):
    case = CASES[name]
    nr, nc = case["nr"], case["nc"]
    n, seed, offset = 160, 99, 1000
    eng = me.MetropolisEngine(case["spec"], None, case["real"], case["cplx"], temp=case["temp"], n_chains=n, seed=seed,
                              dtype="f64", chain_offset=offset, complex_sample_method=case["method"])
    ora = ManyChainOracle(nr, nc, case["energy"], n, seed=seed, temp=case["temp"], initial_real_params=case["real"],
                          initial_complex_params=case["cplx"], chain_offset=offset)
    magphase = case["method"] == "magnitude-phase"
    for cycle in range(case["cycles"]):
        for op in case["ops"]:
            if op == "measure":
                eng.measure()
                ora.measure()
            elif op == "all":
                eng.step_all()
                ora.step(1, group="all")
            elif op == "real":
                eng.step_real_group()
                ora.step(1, group="real")
            elif magphase:
                assert eng.step_complex_group() is None
                ora.step_magnitude_phase()
            else:
                eng.step_complex_group()
                ora.step(1, group="complex")
        if cycle % 10 == 9 or cycle == case["cycles"] - 1:
            assert np.allclose(eng._get(0), ora.x, rtol=0, atol=TOL), "state differs in cycle %d" % cycle
            assert np.allclose(eng.energy_total, ora.energy, rtol=0, atol=TOL)
            if nr:
                assert np.allclose(eng.real_group_sampling_width, ora.width_real, rtol=0, atol=TOL)
            assert np.allclose(eng.complex_group_sampling_width, ora.width_complex, rtol=0, atol=TOL)
            if nr and nc:
                # sampling_width (row 0) only moves in step_all; after one it equals both group widths (:436-437)
                assert np.allclose(eng.sampling_width, ora.width_all, rtol=0, atol=TOL)
            assert np.allclose(eng._get(3), ora.mean, rtol=0, atol=TOL)
            assert np.allclose(eng.covariance_matrix_complex, ora.cov_complex, rtol=0, atol=TOL)
    assert eng.accept_stats() == (ora.accepted, ora.proposed)


# oracle/scenarios.py:_coupled as a dense quadratic form over [x0, x1, Re z0, Re z1, Im z0, Im z1]
_COUPLED = np.zeros((6, 6))
_COUPLED[0, 0], _COUPLED[1, 1] = 1.0, 2.0
_COUPLED[0, 1] = _COUPLED[1, 0] = 0.3
_COUPLED[2, 2] = _COUPLED[4, 4] = 1.5
_COUPLED[3, 3] = _COUPLED[5, 5] = 3.0
_COUPLED[2, 3] = _COUPLED[3, 2] = 0.4
_COUPLED[4, 5] = _COUPLED[5, 4] = 0.4
_COUPLED[4, 3] = _COUPLED[3, 4] = 0.7
_COUPLED[2, 5] = _COUPLED[5, 2] = -0.7

GOLDEN = {
    "groups_landau_terms": (me.LandauToy(1.0, -1.0, 0.5), "multivariate-gaussian"),
    "magphase_1real_2complex": (me.DiagQuadratic((0.5,), (1.0, 3.0)), "magnitude-phase"),
    "magphase_2complex": (me.DiagQuadratic((), (1.0, 2.0)), "magnitude-phase"),
}


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_reference_golden_step_kinds_on_gpu(name, golden_dir):
    """tests/golden/traj_*.npz recorded from the reference with step_real_group / step_complex_group called directly
    (and complex_sample_method="magnitude-phase"), replayed through the HIP kernels in float64."""
    spec = scenarios.SCENARIOS[name]
    gold = np.load(os.path.join(golden_dir, "traj_%s.npz" % name))
    energy_spec, method = GOLDEN[name]
    nr, nc = scenarios.dims(spec)
    eng = me.MetropolisEngine(energy_spec, None, spec["real"], spec["cplx"], temp=spec["temp"], n_chains=1, dtype="f64",
                              complex_sample_method=method)
    t = 0
    for k in range(spec["n_measures"]):
        for op in scenarios.ops(spec)[:-1]:
            if op == "complex" and method == "magnitude-phase":
                eng.step_injected(gold["normals"][t:t + 1, None, :nc], gold["uniforms"][t:t + 1, None, :],
                                  kind=_capi.STEP_COMPLEX_MAGNITUDE_PHASE)
            else:
                kind = {"all": _capi.STEP_ALL, "real": _capi.STEP_REAL_GROUP, "complex": _capi.STEP_COMPLEX_GROUP}[op]
                eng.step_injected(gold["normals"][t:t + 1, None, :], gold["uniforms"][t:t + 1, :1], kind=kind)
            assert np.allclose(eng.real_params, gold["real_params"][t], rtol=0, atol=TOL), (t, op)
            assert np.allclose(eng.complex_params, gold["complex_params"][t], rtol=0, atol=TOL), (t, op)
            assert abs(eng.real_group_sampling_width - gold["real_width"][t]) < TOL
            assert abs(eng.complex_group_sampling_width - gold["complex_width"][t]) < TOL
            t += 1
        eng.measure()
        assert np.allclose(eng.real_mean, gold["real_mean"][k], rtol=0, atol=TOL)
        assert np.allclose(eng.complex_mean, gold["complex_mean"][k], rtol=0, atol=TOL)
        if nr:
            assert np.allclose(eng.covariance_matrix_real, gold["cov_real"][k], rtol=0, atol=TOL)
        assert np.allclose(eng.covariance_matrix_complex, gold["cov_complex"][k], rtol=0, atol=TOL)
        assert np.allclose(eng.observables_mean, gold["observables_mean"][k], rtol=0, atol=TOL)


def test_reference_golden_mixed_call_styles_stale_ledgers(golden_dir):
    """groups_2real_2complex: (real group, complex group, step_all, real group, measure) x 120 recorded from the reference.
    Its step_all decides against `energy_total`, which the group steps never update, and its group steps against
    `energy[term]`, which step_all never updates (metropolis_engine.py:252-255, :214-221, :230-237): every decision,
    both ledgers, widths and running statistics are reproduced with reference_energy_ledgers=True."""
    name = "groups_2real_2complex"
    spec = scenarios.SCENARIOS[name]
    gold = np.load(os.path.join(golden_dir, "traj_%s.npz" % name))
    nr, nc = scenarios.dims(spec)
    eng = me.MetropolisEngine(me.DenseQuadratic(_COUPLED), None, spec["real"], spec["cplx"], temp=spec["temp"], n_chains=1,
                              dtype="f64", reference_energy_ledgers=True)
    t, stale_seen = 0, False
    for k in range(spec["n_measures"]):
        for op in scenarios.ops(spec)[:-1]:
            kind = {"all": _capi.STEP_ALL, "real": _capi.STEP_REAL_GROUP, "complex": _capi.STEP_COMPLEX_GROUP}[op]
            eng.step_injected(gold["normals"][t:t + 1, None, :], gold["uniforms"][t:t + 1, :1], kind=kind)
            assert np.allclose(eng.real_params, gold["real_params"][t], rtol=0, atol=TOL), (t, op)
            assert np.allclose(eng.complex_params, gold["complex_params"][t], rtol=0, atol=TOL), (t, op)
            assert abs(eng.real_group_sampling_width - gold["real_width"][t]) < TOL
            assert abs(eng.complex_group_sampling_width - gold["complex_width"][t]) < TOL
            assert abs(eng.energy_total - gold["energy_total"][t]) < TOL, (t, op)
            assert abs(eng.energy["total"] - gold["energy_terms"][t, 0]) < TOL, (t, op)
            stale_seen |= abs(gold["energy_total"][t] - gold["energy_terms"][t, 0]) > 1e-6
            t += 1
        eng.measure()
        assert np.allclose(eng.real_mean, gold["real_mean"][k], rtol=0, atol=TOL)
        assert np.allclose(eng.covariance_matrix_real, gold["cov_real"][k], rtol=0, atol=TOL)
        assert np.allclose(eng.covariance_matrix_complex, gold["cov_complex"][k], rtol=0, atol=TOL)
    assert stale_seen                      # the scenario really exercises the quirk
    # pure engines have no separate step_all ledger: the flag is refused there
    with pytest.raises(ValueError):
        me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0, 0.0], None, temp=1.0, n_chains=4, reference_energy_ledgers=True)


def test_f32_group_stepping_is_a_valid_sampler():
    """Alternating group steps (the author's cylinder driving pattern) and the magnitude-phase pair leave the
    Boltzmann distribution invariant: Var x = T/(2a), E|z|^2 = T/b (anchor 3)."""
    n = 1 << 15
    a, b, temp = (1.0, 4.0), (2.0, 0.5), 0.5
    for method in ("multivariate-gaussian", "magnitude-phase"):
        eng = me.MetropolisEngine(me.DiagQuadratic(a, b), None, [0.0, 0.0], [0.1 + 0j, 0.1j], temp=temp, n_chains=n,
                                  seed=17, sampling_width=0.4, complex_sample_method=method)
        for _ in range(80):
            eng.step_real_group(5)
            eng.step_complex_group(5)
            eng.measure()
        st = moments_to_statistics(eng.pooled_moments(), 2, 2)
        var = np.diag(st["covariance"])
        # The reference's magnitude stage accepts with exp(-dE/T) only -- no |z| Jacobian for the radial move -- so
        # its invariant density is exp(-E/T)/|z| per complex parameter (radius half-normal): E|z|^2 = T/(2b), half
        # the Boltzmann value.  Reproduced faithfully here (reference quirk).
        cz = 4.0 if method == "magnitude-phase" else 2.0
        want = np.array([temp / (2 * a[0]), temp / (2 * a[1]), temp / (cz * b[0]), temp / (cz * b[1]),
                         temp / (cz * b[0]), temp / (cz * b[1])])
        assert np.all(np.abs(var / want - 1) < 6 * np.sqrt(2.0 / n)), (method, var / want)
        widths = eng._get(2)
        assert widths.shape == (n, 3) and not np.allclose(widths[:, 1], widths[:, 2])   # the group widths decouple


def test_shared_factor_mixed_engine_follows_the_oracle():
    """cov_mode="pooled" on a mixed engine: ONE proposal shape for all chains, a real Cholesky factor and the factor of
    conj(K) for the complex block (metropolis_engine.py:261-302 with the quirk-Q3 convention), installed with
    set_shared_factor; step_all and group steps follow the oracle run with the same fixed covariance matrices."""
    from metropolisengine_amd.distributed import pooled_factor
    nr, nc, n, seed = 2, 2, 150, 71
    rng = np.random.default_rng(12)
    br = rng.standard_normal((nr, nr))
    c_r = br @ br.T + 0.5 * np.identity(nr)
    bc = rng.standard_normal((nc, nc)) + 1j * rng.standard_normal((nc, nc))
    k_c = bc @ bc.conj().T + 0.5 * np.identity(nc)                      # Hermitian positive definite, off-diagonal phase
    cov = np.zeros((nr + 2 * nc, nr + 2 * nc))                          # real representation of (c_r, k_c)
    a, b = slice(nr, nr + nc), slice(nr + nc, nr + 2 * nc)
    cov[:nr, :nr] = c_r
    cov[a, a] = cov[b, b] = k_c.real / 2
    cov[b, a] = k_c.imag / 2
    cov[a, b] = -k_c.imag / 2
    spec = me.DiagQuadratic((1.0, 2.0), (1.5, 3.0))
    eng = me.MetropolisEngine(spec, None, [0.1, -0.1], [0.2 + 0.1j, -0.1 + 0.3j], temp=1.0, n_chains=n, seed=seed, dtype="f64",
                              cov_mode="pooled", sampling_width=0.2)
    eng.set_shared_factor(pooled_factor(cov, nr, nc))
    ora = ManyChainOracle(nr, nc, energies.diag_quadratic(nr, nc, (1.0, 2.0), (1.5, 3.0)), n, seed=seed, temp=1.0,
                          initial_real_params=[0.1, -0.1], initial_complex_params=[0.2 + 0.1j, -0.1 + 0.3j],
                          sampling_width=0.2, covariance_matrix_real=c_r, covariance_matrix_complex=k_c, adapt_shape=False)
    for cycle in range(25):
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
    assert np.allclose(eng._get(0), ora.x, rtol=0, atol=TOL)
    assert np.allclose(eng.energy_total, ora.energy, rtol=0, atol=TOL)
    assert np.allclose(eng.real_group_sampling_width, ora.width_real, rtol=0, atol=TOL)
    assert np.allclose(eng.complex_group_sampling_width, ora.width_complex, rtol=0, atol=TOL)
    assert np.allclose(eng._get(3), ora.mean, rtol=0, atol=TOL)
    assert eng.accept_stats() == (ora.accepted, ora.proposed)
    assert 0.05 < ora.accepted / ora.proposed < 0.9
