"""The plain-C restatement (oracle/c/me_oracle.c: checker + strong CPU baseline) against the numpy many-chain oracle
on identical Philox streams: float64 both, differing only in libm vs numpy transcendentals (tolerance 1e-12)."""
import numpy as np
import pytest

from oracle import energies
from oracle.c_oracle import COracle
from oracle.manychain import ManyChainOracle

CASES = {
    "config2_16real": dict(nr=16, nc=0, a=[1.0] * 16, b=[], real=[0.0] * 16, cplx=None, temp=1.0),
    "config3_4real_4complex": dict(nr=4, nc=4, a=[1, 2, 4, 8], b=[1, 2, 4, 8], real=[0.1, 0.2, -0.1, 0.0],
                                   cplx=[0.1j, 0.2, -0.1 + 0.1j, 0.0], temp=1.0),
    "readme_1real": dict(nr=1, nc=0, a=[1.0], b=[], real=[0.0], cplx=None, temp=0.01),
    "pure_complex_zero_temp": dict(nr=0, nc=3, a=[], b=[1.0, 2.0, 0.5], real=None, cplx=[0.4, 0.3j, -0.2], temp=0.0),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_c_oracle_matches_numpy_oracle(name):
    case = CASES[name]
    nr, nc = case["nr"], case["nc"]
    n, seed, offset = 200, 2026, (1 << 33) + 5
    c = COracle(nr, nc, case["a"], case["b"], n_chains=n, seed=seed, temp=case["temp"],
                initial_real_params=case["real"], initial_complex_params=case["cplx"], chain_offset=offset)
    py = ManyChainOracle(nr, nc, energies.diag_quadratic(nr, nc, case["a"], case["b"]), n, seed=seed, temp=case["temp"],
                         initial_real_params=case["real"], initial_complex_params=case["cplx"], chain_offset=offset)
    for _ in range(6):
        c.step(25)
        py.step(25)
        assert np.allclose(c.x, py.x, rtol=0, atol=1e-12)
        assert np.allclose(c.width, py.width_real if nr else py.width_complex, rtol=0, atol=1e-12)
        assert np.allclose(c.energy, py.energy, rtol=0, atol=1e-12)
    assert (c.accepted, c.proposed) == (py.accepted, py.proposed)
