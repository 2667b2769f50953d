"""GPU parity at scale against the plain-C oracle (oracle/c/me_oracle.c): 2^16 chains of BASELINE config 2 and 3 shapes,
every chain compared (float64, 1e-9); the float32 engine is compared through pooled moments."""
import numpy as np
import pytest

import metropolisengine_amd as me
from oracle.c_oracle import COracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", ["config2", "config3"])
def test_f64_matches_c_oracle_on_65536_chains(shape):
    n, seed, sweeps = 1 << 16, 2026, 120
    if shape == "config2":
        nr, nc, a, b, real0, cplx0 = 16, 0, [1.0] * 16, [], [0.0] * 16, None
        spec = me.IsoQuadratic(1.0)
    else:
        nr, nc, a, b = 4, 4, [1.0, 2.0, 4.0, 8.0], [1.0, 2.0, 4.0, 8.0]
        real0, cplx0 = [0.1, 0.2, -0.1, 0.0], [0.1j, 0.2, -0.1 + 0.1j, 0.0]
        spec = me.DiagQuadratic(a, b)
    eng = me.MetropolisEngine(spec, None, real0, cplx0, temp=1.0, n_chains=n, seed=seed, dtype="f64")
    ref = COracle(nr, nc, a, b, n_chains=n, seed=seed, temp=1.0, initial_real_params=real0, initial_complex_params=cplx0)
    eng.step_all(sweeps)
    ref.step(sweeps)
    assert np.allclose(eng._get(0), ref.x, rtol=0, atol=1e-9)
    assert np.allclose(eng._get(2)[:, 0], ref.width, rtol=0, atol=1e-9)
    assert np.allclose(eng.energy_total, ref.energy, rtol=0, atol=1e-9)
    assert eng.accept_stats() == (ref.accepted, ref.proposed)
    # float32 engine, same seed: different rounding, same distribution
    f32 = me.MetropolisEngine(spec, None, real0, cplx0, temp=1.0, n_chains=n, seed=seed)
    f32.step_all(sweeps)
    acc32, prop32 = f32.accept_stats()
    assert abs(acc32 / prop32 - ref.accepted / ref.proposed) < 2e-3
    x32 = f32._get(0)
    assert np.all(np.abs(x32.var(axis=0) / ref.x.var(axis=0) - 1) < 0.05)
