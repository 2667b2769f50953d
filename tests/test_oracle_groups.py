"""Group-wise stepping and the magnitude-phase sampler in the many-chain oracle, tied to the golden-pinned
single-chain restatement on identical Philox words (tolerance 1e-10, float64 round-off only)."""
import numpy as np
import pytest

from oracle import energies, philox
from oracle.manychain import ManyChainOracle
from oracle.reference_chain import ReferenceChain, StreamSources

TOL = 1e-10


def _streams(seed, chain_id, total, nr, nc, magphase_steps):
    """Per-step normals rows and uniforms rows [accept, phases..., accept2] for one chain, from the Philox spec."""
    d = nr + 2 * nc
    normals = np.zeros((total, d))
    uniforms = np.zeros((total, nc + 2))
    cid = np.array([chain_id], dtype=np.uint64)
    for t in range(total):
        if t in magphase_steps:
            w1 = philox.n_normal_words(nc)
            unit = philox.unit_open(philox.step_words(seed, cid, t, w1 + nc + 2))[0]
            r = np.sqrt(-2.0 * np.log(unit[0:w1:2]))
            th = 2.0 * np.pi * unit[1:w1:2]
            g = np.empty(w1)
            g[0::2], g[1::2] = r * np.cos(th), r * np.sin(th)
            normals[t, :nc] = g[:nc]
            uniforms[t, 0] = unit[w1]
            uniforms[t, 1:1 + nc] = unit[w1 + 1:w1 + 1 + nc]
            uniforms[t, nc + 1] = unit[w1 + nc + 1]
        else:
            g, u = philox.step_draws(seed, cid, t, d)
            normals[t] = g[0]
            uniforms[t, 0] = u[0]
    return normals, uniforms


CASES = {
    "groups_mixed": dict(nr=2, nc=2, method="multivariate-gaussian", ops=("real", "complex", "real", "measure"),
                         energy=energies.diag_quadratic(2, 2, (1.0, 2.0), (1.5, 3.0)), temp=1.0,
                         real=[0.1, -0.1], cplx=[0.2 + 0.1j, -0.1 + 0.3j], cycles=70),
    "magphase_mixed": dict(nr=1, nc=2, method="magnitude-phase", ops=("real", "complex", "measure"),
                           energy=energies.diag_quadratic(1, 2, (0.5,), (1.0, 3.0)), temp=0.5,
                           real=[0.2], cplx=[0.3 + 0.1j, 0.2j], cycles=80),
    "magphase_pure_complex": dict(nr=0, nc=3, method="magnitude-phase", ops=("complex", "all", "complex", "measure"),
                                  energy=energies.diag_quadratic(0, 3, (), (1.0, 2.0, 0.5)), temp=0.5,
                                  real=None, cplx=[0.3 + 0.1j, 0.2j, 0.0], cycles=70),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_group_and_magnitude_phase_steps(name):
    case = CASES[name]
    nr, nc = case["nr"], case["nc"]
    n_chains, seed, offset = 3, 99, 1000
    step_ops = [op for op in case["ops"] if op != "measure"]
    total = len(step_ops) * case["cycles"]
    magphase = case["method"] == "magnitude-phase"
    mag_steps = {t for t in range(total) if magphase and step_ops[t % len(step_ops)] == "complex"}
    many = ManyChainOracle(nr, nc, case["energy"], n_chains, seed=seed, temp=case["temp"],
                           initial_real_params=case["real"], initial_complex_params=case["cplx"], chain_offset=offset)
    singles = []
    for c in range(n_chains):
        normals, uniforms = _streams(seed, offset + c, total, nr, nc, mag_steps)
        singles.append(ReferenceChain(energies.as_reference_callable(case["energy"], nr, nc),
                                      initial_real_params=case["real"], initial_complex_params=case["cplx"],
                                      temp=case["temp"], sources=StreamSources(normals, uniforms, nr, nc),
                                      complex_sample_method=case["method"]))
    for _ in range(case["cycles"]):
        for op in case["ops"]:
            if op == "measure":
                many.measure()
            elif op == "complex" and magphase:
                many.step_magnitude_phase()
            else:
                many.step(1, group=op)
            for c, single in enumerate(singles):
                {"all": single.step_all, "real": single.step_real_group, "complex": single.step_complex_group,
                 "measure": single.measure}[op]()
                assert np.allclose(single.real_params, many.x[c, :nr], rtol=0, atol=TOL)
                assert np.allclose(single.complex_params, many.complex_params()[c], rtol=0, atol=TOL)
                assert abs(single.real_group_sampling_width - many.width_real[c]) < TOL
                assert abs(single.complex_group_sampling_width - many.width_complex[c]) < TOL
                if nr and nc:
                    assert abs(single.sampling_width - many.width_all[c]) < TOL
    for c, single in enumerate(singles):
        if nr:
            assert np.allclose(single.covariance_matrix_real, many.cov_real[c], rtol=0, atol=TOL)
        assert np.allclose(single.covariance_matrix_complex, many.cov_complex[c], rtol=0, atol=TOL)
        assert np.allclose(single.observables_mean, many.observables_mean[c], rtol=0, atol=TOL)
    assert many.accepted == sum(s.accepted for s in singles) and many.proposed == sum(s.proposed for s in singles)
