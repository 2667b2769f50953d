"""Size-independent properties at BASELINE.json's full sizes (configs 3, 4, 5; config 2 is in test_gpu_parity.py):
shard equivalence (chains are independent and their Philox streams are addressed by global id, so any sub-range of a
full-size engine equals a small engine that owns the same ids -- bitwise in float32), stationary statistics against
the analytic values, and the invariants of the protocol (hard wall, finite energies, counters)."""
import os

import numpy as np
import pytest

import metropolisengine_amd as me
from metropolisengine_amd import _capi
from metropolisengine_amd.distributed import moments_to_statistics

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cycle(eng, cycles, steps):
    for _ in range(cycles):
        for _ in range(steps):
            eng.step_all()
        eng.measure()


def test_config2_full_size_float64_headline():
    """The headline workload in the reference's arithmetic: 2^20 chains x 16 real parameters, float64.  Any sub-range is
    bitwise the small engine that owns the same global chain ids (so the 1e-9 oracle parity of the small sizes carries
    over chain by chain), launch splitting does not change results, and the stationary variance is T / (2 a)."""
    n, sub, off = 1 << 20, 200, (1 << 20) - 64 * 99 - 13
    kw = dict(temp=1.0, seed=2026, dtype="f64")
    full = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 16, None, n_chains=n, **kw)
    part = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 16, None, n_chains=sub, chain_offset=off, **kw)
    full.step_all(64)                           # fused sweeps ...
    for _ in range(8):                          # ... against one-sweep and short fused launches of the same 64 steps
        part.step_all()
        part.step_all(7)
    for field in (_capi.FIELD_PARAMS, _capi.FIELD_ENERGY, _capi.FIELD_WIDTH):
        assert np.array_equal(full._get(field, off, sub), part._get(field)), field
    full.step_all(2500)
    st = moments_to_statistics(full.pooled_moments(), 16, 0)
    assert np.all(np.abs(np.diag(st["covariance"]) * 2.0 - 1.0) < 0.01 + 6 * np.sqrt(2.0 / n))
    assert np.all(np.abs(st["mean"]) < 6 * np.sqrt(0.5 / n) + 1e-3)
    assert 0.2 < st["acceptance_rate"] < 0.45
    x = full._get(_capi.FIELD_PARAMS, 0, 4096)
    assert np.allclose(full._get(_capi.FIELD_ENERGY, 0, 4096)[:, 0], (x * x).sum(axis=1), rtol=1e-12)


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_config3_full_size_adaptive_covariance(dtype):
    n, sub, off = 1 << 20, 192, (1 << 20) - 4096 - 37
    a = b = (1.0, 2.0, 4.0, 8.0)
    kw = dict(temp=1.0, seed=2026, dtype=dtype)
    full = me.MetropolisEngine(me.DiagQuadratic(a, b), None, [0.0] * 4, [0j] * 4, n_chains=n, **kw)
    part = me.MetropolisEngine(me.DiagQuadratic(a, b), None, [0.0] * 4, [0j] * 4, n_chains=sub, chain_offset=off, **kw)
    for eng in (full, part):
        _cycle(eng, 70, 10)                    # > 50 measures: every chain proposes with its own covariance
    for field in (_capi.FIELD_PARAMS, _capi.FIELD_ENERGY, _capi.FIELD_WIDTH, _capi.FIELD_MEAN, _capi.FIELD_COV,
                  _capi.FIELD_FACTOR, _capi.FIELD_OBS_MEAN):
        assert np.array_equal(full._get(field, off, sub), part._get(field)), field
    assert full.measure_step_counter == 71
    full.step_all(3000)                         # the a = 1 mode equilibrates from the all-zero start slowest
    st = moments_to_statistics(full.pooled_moments(), 4, 4)
    se = 6 * np.sqrt(2.0 / n)
    var = np.diag(st["covariance"])
    assert np.all(np.abs(var[:4] * 2 * np.array(a) - 1.0) < 0.02 + se)          # var x_i = T / (2 a_i)
    mod2 = var[4:8] + var[8:12]
    assert np.all(np.abs(mod2 * np.array(b) - 1.0) < 0.02 + se)                 # E|z_j|^2 = T / b_j
    assert 0.2 < st["acceptance_rate"] < 0.5


def test_config4_full_size_dense_matrix_cores():
    n, sub = 1 << 19, 128
    m = np.random.default_rng(5).standard_normal((64, 64))
    amat = m @ m.T / 64 + np.identity(64)
    kw = dict(temp=1.0, seed=2026, sampling_width=0.15, cov_mode="fixed")
    full = me.MetropolisEngine(me.DenseQuadratic(amat), None, [0.0] * 64, None, n_chains=n, **kw)
    part = me.MetropolisEngine(me.DenseQuadratic(amat), None, [0.0] * 64, None, n_chains=sub, chain_offset=n - sub, **kw)
    for eng in (full, part):
        eng.step_all(150)
        for _ in range(50):
            eng.step_all()
    assert np.array_equal(full._get(_capi.FIELD_PARAMS, n - sub, sub), part._get(_capi.FIELD_PARAMS))
    assert np.array_equal(full._get(_capi.FIELD_ENERGY, n - sub, sub), part._get(_capi.FIELD_ENERGY))
    x = full._get(_capi.FIELD_PARAMS, 0, 2048)
    e = full._get(_capi.FIELD_ENERGY, 0, 2048)[:, 0]
    assert np.allclose(e, np.einsum("ni,ij,nj->n", x, amat, x), rtol=2e-5)      # the ledger is the energy of the state
    full.step_all(1500)
    st = moments_to_statistics(full.pooled_moments(), 64, 0)
    want = 0.5 * np.linalg.inv(amat)
    assert np.all(np.abs(st["covariance"] - want) < 0.02 * np.max(np.abs(want)) + 6 * np.max(np.abs(want)) * np.sqrt(2.0 / n))
    acc, prop = full.accept_stats()
    assert prop == n * 1700 and 0.1 < acc / prop < 0.6


def test_config5_full_size_user_energy_and_wall():
    n, sub, off = 1 << 18, 256, 77 * 64 + 3
    src = os.path.join(REPO, "examples", "user_energy_cylinder.h")
    def make(count, offset):
        return me.MetropolisEngine(me.UserEnergy("cylinder", src, (1.0, 0.5, 1.0)), me.AbsReal0AtLeast(1.0), [0.1, 0.0],
                                   [0.05] * 7, temp=0.1, n_chains=count, chain_offset=offset, seed=2026)
    full, part = make(n, 0), make(sub, off)
    for eng in (full, part):
        _cycle(eng, 60, 10)
    for field in (_capi.FIELD_PARAMS, _capi.FIELD_ENERGY, _capi.FIELD_WIDTH, _capi.FIELD_COV, _capi.FIELD_FACTOR):
        assert np.array_equal(full._get(field, off, sub), part._get(field)), field
    x = full._get(_capi.FIELD_PARAMS)
    assert np.all(np.abs(x[:, 0]) < 1.0)                     # no chain ever crossed the hard wall
    assert np.all(np.isfinite(full._get(_capi.FIELD_ENERGY)))
    st = moments_to_statistics(full.pooled_moments(), 2, 7)
    assert st["n_chains"] == n and 0.15 < st["acceptance_rate"] < 0.5
    # the field modes are circular: <Re z Im z> = 0 and <Re z^2> = <Im z^2> within the ensemble error
    cov = st["covariance"]
    re, im = slice(2, 9), slice(9, 16)
    scale = np.max(np.diag(cov)[2:])
    assert np.all(np.abs(np.diag(cov[re, im])) < 8 * scale * np.sqrt(1.0 / n) + 1e-4)
    assert np.all(np.abs(np.diag(cov)[re] - np.diag(cov)[im]) < 8 * scale * np.sqrt(2.0 / n) + 1e-4)


def test_engines_beyond_the_4gib_packed_field_keep_no_per_chain_covariance():
    """16 real parameters x 2^22 chains in float64: the packed covariance field would be 4.6 GB, past what one buffer
    descriptor spans.  cov_mode="reference" is refused with a clear message; the fixed / pooled shapes run (this is the
    engine behind bench.py's `roofline_hbm`) and simply keep no per-chain covariance."""
    n = 1 << 22
    with pytest.raises(NotImplementedError, match="4 GiB"):
        me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 16, None, temp=1.0, n_chains=n, seed=1, dtype="f64")
    eng = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 16, None, temp=1.0, n_chains=n, seed=1, dtype="f64",
                              cov_mode="fixed")
    eng.step_all(60)
    for _ in range(3):
        eng.step_all()
        eng.measure()
    with pytest.raises(NotImplementedError):
        eng.covariance_matrix_real
    st = moments_to_statistics(eng.pooled_moments(), 16, 0)
    assert st["n_chains"] == n and 0.2 < st["acceptance_rate"] < 0.95
    mean = eng._get(_capi.FIELD_MEAN, n - 100, 100)
    assert np.all(np.isfinite(mean)) and np.any(mean != 0.0)
