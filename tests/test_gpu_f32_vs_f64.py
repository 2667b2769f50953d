"""The float32 (production) kernels against the float64 kernels on the SAME Philox streams, many sweeps deep.

The float64 instantiation is the one pinned to the oracle and the reference's goldens at 1e-9; this file ties the float32
instantiation to it: from a common state both run the same sweeps, and a chain's ACCEPT SEQUENCE (did the state change
at sweep k?) must be identical for >= 99 % of the chains -- the rest are near-ties of u against exp(-dE/T) that a 1e-7
difference in dE flips, after which a chain legitimately follows another path.  For the chains with identical
sequences the state agrees to 1e-4 and the adapted width to 1e-5 relative.  Covered: config 2 (16 real, identity shape),
config 3 (4 real + 4 complex with per-chain adaptive factors active, i.e. after more than 50 measures) and config 5's
shape (2 real + 7 complex, user plugin, hard wall).  Last, a 2 000-step comparison of the ensemble-mean width catches a
drift between the two width recursions (Num<float>::adapt multiplies by a precomputed factor, Num<double>::adapt
follows the literal order of metropolis_engine.py:431-435)."""
import os

import numpy as np
import pytest

import metropolisengine_amd as me

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pair(make, n_chains, warm):
    """A float64 engine run through `warm`, and a float32 engine started from its state."""
    f64 = make("f64", n_chains)
    warm(f64)
    f32 = make("f32", n_chains)
    f32.load_state_dict(f64.state_dict())
    f64.load_state_dict(f32.state_dict())      # both now hold the float32-representable state exactly
    for field in (0, 2):
        assert np.array_equal(f32._get(field), f64._get(field))
    f32.initialize_energy_dict()
    f64.initialize_energy_dict()
    return f32, f64


def _compare(f32, f64, sweeps, measure_every=0):
    n = f32.n_chains
    same = np.ones(n, dtype=bool)
    x32, x64 = f32._get(0), f64._get(0)
    for k in range(sweeps):
        f32.step_all()
        f64.step_all()
        if measure_every and (k + 1) % measure_every == 0:
            f32.measure()
            f64.measure()
        y32, y64 = f32._get(0), f64._get(0)
        same &= np.any(y32 != x32, axis=1) == np.any(y64 != x64, axis=1)
        x32, x64 = y32, y64
    assert same.mean() >= 0.99, same.mean()
    scale = np.maximum(1.0, np.abs(x64[same]))
    assert np.max(np.abs(x32[same] - x64[same]) / scale) < 1e-4
    w32, w64 = f32._get(2)[same], f64._get(2)[same]
    assert np.max(np.abs(w32 - w64) / w64) < 1e-5
    e32, e64 = f32._get(1)[same], f64._get(1)[same]
    assert np.max(np.abs(e32 - e64) / np.maximum(1.0, np.abs(e64))) < 1e-4
    return same


def test_config2_identity_shape_50_sweeps():
    def make(dtype, n):
        return me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 16, None, temp=1.0, n_chains=n, seed=2026, dtype=dtype)
    f32, f64 = _pair(make, 8192, lambda e: e.step_all(300))
    same = _compare(f32, f64, 50)
    assert same.mean() > 0.995


def test_config3_adaptive_factors_50_sweeps():
    a = (1.0, 2.0, 4.0, 8.0)

    def make(dtype, n):
        return me.MetropolisEngine(me.DiagQuadratic(a, a), None, [0.0] * 4, [0j] * 4, temp=1.0, n_chains=n, seed=2026, dtype=dtype)

    def warm(e):
        for _ in range(70):                       # > 50 measures: per-chain covariance and Cholesky factors are live
            e.step_all(10)
            e.measure()
    f32, f64 = _pair(make, 4096, warm)
    assert f32.measure_step_counter == f64.measure_step_counter == 71
    # the factors both engines propose with come from the SAME float64 checkpoint
    assert np.allclose(f32._get(6), f64._get(6), rtol=1e-6, atol=1e-7)
    same = _compare(f32, f64, 50, measure_every=10)      # measures in between refresh the factors in each precision
    # running covariance and refreshed factors of the chains that took the same path
    assert np.allclose(f32._get(4)[same], f64._get(4)[same], rtol=1e-4, atol=1e-5)
    assert np.allclose(f32._get(6)[same], f64._get(6)[same], rtol=1e-4, atol=1e-5)


def test_config5_user_plugin_and_wall_50_sweeps():
    src = os.path.join(ROOT, "examples", "user_energy_cylinder.h")

    def make(dtype, n):
        return me.MetropolisEngine(me.UserEnergy("cylinder", src, (1.0, 0.5, 1.0)), me.AbsReal0AtLeast(1.0), [0.1, 0.0],
                                   [0.05] * 7, temp=0.1, n_chains=n, seed=2026, dtype=dtype, cov_mode="fixed")
    f32, f64 = _pair(make, 4096, lambda e: e.step_all(400))
    _compare(f32, f64, 50)
    assert np.all(np.abs(f32._get(0)[:, 0]) < 1.0)


@pytest.mark.parametrize("mode", ["fixed", "pooled"], ids=["identity-shape", "shared-factor"])
def test_config4_dense64_matrix_core_kernels_50_sweeps(mode):
    """Config 4: the float32 matrix-core kernel (split-bf16, me_dense_bf16x3.h) against the float64 one
    (v_mfma_f64_16x16x4_f64, me_dense_f64.h, itself pinned to the oracle as a trajectory in test_gpu_dense_f64.py):
    identical accept sequences over 50 sweeps, identity shape and the shared factor L g on the matrix cores."""
    m = np.random.default_rng(5).standard_normal((64, 64))
    amat = m @ m.T / 64 + np.identity(64)

    def make(dtype, n):
        eng = me.MetropolisEngine(me.DenseQuadratic(amat), None, [0.0] * 64, None, temp=1.0, n_chains=n, seed=2026, dtype=dtype,
                                  cov_mode=mode, sampling_width=0.2)
        if mode == "pooled":
            b = np.random.default_rng(9).standard_normal((64, 64))
            cov = 0.5 * np.linalg.inv(amat) + 0.02 * (b @ b.T) / 64
            chol = np.linalg.cholesky(cov)
            eng.set_shared_factor(chol[np.tril_indices(64)])
        return eng
    f32, f64 = _pair(make, 4096, lambda e: e.step_all(200))
    same = _compare(f32, f64, 50)
    assert same.mean() > 0.99


def test_width_recursion_does_not_drift_over_2000_steps():
    """Ensemble-mean width of float32 against float64 every 100 steps (both adapt from 0.05 towards ~0.56 at 16
    parameters): equal within the ensemble error plus 1e-3 relative, at every checkpoint."""
    n = 8192
    kw = dict(temp=1.0, n_chains=n, seed=77)
    f32 = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 16, None, dtype="f32", **kw)
    f64 = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 16, None, dtype="f64", **kw)
    worst = 0.0
    for _ in range(20):
        f32.step_all(100)
        f64.step_all(100)
        w32, w64 = f32._get(2)[:, 0], f64._get(2)[:, 0]
        se = np.hypot(w32.std(), w64.std()) / np.sqrt(n)
        diff = abs(w32.mean() - w64.mean())
        assert diff < 5 * se + 1e-3 * w64.mean(), (diff, se)
        worst = max(worst, diff / w64.mean())
    assert 0.3 < w64.mean() < 0.9 and worst < 5e-3
    # and the per-step multiplicative update itself: a chain that accepts everything / nothing for 200 steps
    ratio, p, damping = f64.ratio, 0.3, 200.0
    up32 = np.float32(1.0) + np.float32(ratio * (1 - p) / damping)
    w = np.float32(0.05)
    wd = 0.05
    for _ in range(200):
        w = np.float32(w * up32)
        wd = wd + wd * ratio * (1 - p) / damping
    assert abs(float(w) / wd - 1) < 2e-5
