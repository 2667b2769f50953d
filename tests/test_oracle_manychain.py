"""Ties oracle/manychain.py (the semantics the HIP kernels implement) to the golden-pinned single-chain restatement.

Chain c of the many-chain oracle must follow ReferenceChain driven by the same Philox words, step for step.
Tolerance 1e-10 absolute: both are float64; they differ only in rounding order (sigma * chol(C) vs chol(sigma^2 C)).
"""
import numpy as np
import pytest

from oracle import energies, philox
from oracle.manychain import ManyChainOracle
from oracle.reference_chain import ReferenceChain, StreamSources

TOL = 1e-10

rng = np.random.default_rng(5)
_M = rng.standard_normal((6, 6))
DENSE6 = _M @ _M.T / 6 + np.identity(6)

CASES = {
    "iso_16real": dict(nr=16, nc=0, energy=energies.iso_quadratic(16, 0, 1.0), temp=1.0,
                       real=[0.0] * 16, cplx=None, spm=2, nm=70),
    "diag_4real_4complex": dict(nr=4, nc=4, energy=energies.diag_quadratic(4, 4, (1, 2, 4, 8), (1, 2, 4, 8)), temp=1.0,
                                real=[0.1, 0.2, -0.1, 0.0], cplx=[0.1j, 0.2, -0.1 + 0.1j, 0.0], spm=3, nm=80),
    "dense_2real_2complex": dict(nr=2, nc=2, energy=energies.dense_quadratic(2, 2, DENSE6), temp=0.7,
                                 real=[0.3, -0.3], cplx=[0.1 + 0.1j, -0.2j], spm=2, nm=90),
    "diag_3complex": dict(nr=0, nc=3, energy=energies.diag_quadratic(0, 3, (), (1, 3, 0.5)), temp=0.5,
                          real=None, cplx=[0.1, 0.1j, -0.1], spm=2, nm=70),
    "landau": dict(nr=2, nc=1, energy=energies.landau_toy(), temp=0.1, real=[0.0, 0.0], cplx=[0j], spm=10, nm=60),
    "wall": dict(nr=1, nc=1, energy=energies.diag_quadratic(1, 1, (0.5,), (1.0,)), temp=1.0, real=[0.0], cplx=[0.1j],
                 spm=3, nm=70, reject=energies.wall_reject(0.25)),
    "zero_temp": dict(nr=2, nc=0, energy=energies.diag_quadratic(2, 0, (1.0, 3.0)), temp=0.0, real=[1.0, -1.0],
                      cplx=None, spm=4, nm=30),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_manychain_follows_reference_chain(name):
    case = CASES[name]
    nr, nc = case["nr"], case["nc"]
    d = nr + 2 * nc
    n_chains, seed, offset = 3, 2026, (1 << 33) + 17
    total = case["spm"] * case["nm"]
    many = ManyChainOracle(nr, nc, case["energy"], n_chains, seed=seed, temp=case["temp"],
                           initial_real_params=case["real"], initial_complex_params=case["cplx"],
                           chain_offset=offset, reject=case.get("reject"))
    singles = []
    for c in range(n_chains):
        cid = np.array([offset + c], dtype=np.uint64)
        draws = [philox.step_draws(seed, cid, t, d) for t in range(total)]
        normals = np.array([g[0] for g, _ in draws])
        uniforms = np.array([u[0] for _, u in draws])
        rej = case.get("reject")
        singles.append(ReferenceChain(
            energies.as_reference_callable(case["energy"], nr, nc), initial_real_params=case["real"],
            initial_complex_params=case["cplx"], temp=case["temp"],
            reject_condition=None if rej is None else energies.as_reference_reject(rej, nr, nc),
            sources=StreamSources(normals, uniforms, nr, nc)))
    for k in range(case["nm"]):
        for _ in range(case["spm"]):
            many.step()
            for c, single in enumerate(singles):
                acc = single.step_all()
                assert acc == bool(many.last_accept[c])
                assert np.allclose(single.real_params, many.x[c, :nr], rtol=0, atol=TOL)
                assert np.allclose(single.complex_params, many.complex_params()[c], rtol=0, atol=TOL)
                assert abs(single.real_group_sampling_width - many.width_real[c]) < TOL
                assert abs(single.complex_group_sampling_width - many.width_complex[c]) < TOL
        many.measure()
        for c, single in enumerate(singles):
            single.measure()
            assert np.allclose(single.real_mean, many.mean[c, :nr], rtol=0, atol=TOL)
            assert np.allclose(single.complex_mean, many.mean[c, nr:nr + nc] + 1j * many.mean[c, nr + nc:],
                               rtol=0, atol=TOL)
            if nr:
                assert np.allclose(single.covariance_matrix_real, many.cov_real[c], rtol=0, atol=TOL)
            if nc:
                assert np.allclose(single.covariance_matrix_complex, many.cov_complex[c], rtol=0, atol=TOL)
            assert np.allclose(single.observables_mean, many.observables_mean[c], rtol=0, atol=TOL)
    assert many.accepted == sum(s.accepted for s in singles)
    assert many.measure_step_counter == case["nm"] + 1 and many.step_index == total


def test_sharding_invariance():
    """Chains are addressed by global id: two half-size shards reproduce the full run bit for bit."""
    kw = dict(nr=2, nc=1, energy=energies.landau_toy(), seed=11, temp=0.1, initial_real_params=[0.0, 0.0],
              initial_complex_params=[0j])
    full = ManyChainOracle(n_chains=8, **kw)
    lo = ManyChainOracle(n_chains=4, chain_offset=0, **kw)
    hi = ManyChainOracle(n_chains=4, chain_offset=4, **kw)
    for k in range(60):
        for o in (full, lo, hi):
            o.step(3)
            o.measure()
    assert np.array_equal(full.x, np.concatenate((lo.x, hi.x)))
    assert np.array_equal(full.cov_real, np.concatenate((lo.cov_real, hi.cov_real)))
    assert np.allclose(full.pooled_moments(), lo.pooled_moments() + hi.pooled_moments(), rtol=1e-13)


def test_stationary_moments_quadratic():
    """Analytic anchors of SURVEY.md section 4 (3): Var x_i = T/(2 a_i), E|z_j|^2 = T/b_j."""
    a, b, temp = (1.0, 4.0), (2.0,), 0.5
    o = ManyChainOracle(2, 1, energies.diag_quadratic(2, 1, a, b), n_chains=4000, seed=3, temp=temp,
                        initial_real_params=[0.0, 0.0], initial_complex_params=[0j], sampling_width=0.5)
    o.step(400)
    var = o.x.var(axis=0)
    se = 4 * np.sqrt(2.0 / 4000)          # 4 sigma relative error of a variance estimate from 4000 samples
    assert abs(var[0] / (temp / (2 * a[0])) - 1) < se
    assert abs(var[1] / (temp / (2 * a[1])) - 1) < se
    assert abs((var[2] + var[3]) / (temp / b[0]) - 1) < se
    assert 0.2 < o.accepted / o.proposed < 0.45
