"""Host-side equilibration statistics (metropolisengine_amd/statistics.py; reference statistics.py:25-64).

PARITY UNPINNED: the reference delegates to pymbar (absent offline, unpinned); these tests check the restated published
algorithm against analytic results only: for an AR(1) process x_t = phi x_{t-1} + e_t the statistical inefficiency is
g = (1 + phi) / (1 - phi)."""
import numpy as np
import pandas
import pytest

from metropolisengine_amd import statistics


def ar1(n, phi, rng, start=0.0):
    x = np.empty(n)
    x[0] = start
    noise = rng.standard_normal(n) * np.sqrt(1 - phi * phi)
    for t in range(1, n):
        x[t] = phi * x[t - 1] + noise[t]
    return x


@pytest.mark.parametrize("phi", [0.0, 0.5, 0.8])
def test_statistical_inefficiency_of_ar1(phi):
    rng = np.random.default_rng(3)
    want = (1 + phi) / (1 - phi)
    got = np.mean([statistics.statistical_inefficiency(ar1(20000, phi, rng)) for _ in range(4)])
    assert abs(got / want - 1) < 0.15
    fast = statistics.statistical_inefficiency(ar1(20000, phi, rng), fast=True)
    assert abs(fast / want - 1) < 0.3


def test_detects_the_end_of_a_transient():
    rng = np.random.default_rng(4)
    series = ar1(3000, 0.5, rng)
    series[:400] += np.linspace(8.0, 0.0, 400)           # relaxation from a far-away start
    t0, g, neff = statistics.detect_equilibration(series)
    assert 200 < t0 < 700 and 1.5 < g < 6 and neff > 300
    assert statistics.detect_equilibration(np.ones(50)) == (0, 1.0, 1.0)
    with pytest.raises(ValueError):
        statistics.statistical_inefficiency(np.ones(10))


def test_frame_level_helpers():
    rng = np.random.default_rng(5)
    n = 1500
    relax = np.concatenate((np.linspace(5, 0, 300), np.zeros(n - 300)))
    df = pandas.DataFrame({"abs_param_0": np.abs(ar1(n, 0.3, rng) + relax),
                           "total_energy": ar1(n, 0.6, rng) + 2 * relax,
                           "param_0": ar1(n, 0.3, rng) + 1j * ar1(n, 0.3, rng),
                           "real_group_sampling_width": np.linspace(0.05, 0.5, n),
                           "constant": np.full(n, 2.0)})
    points = statistics.get_equilibration_points(df)
    assert set(points) == {"abs_param_0", "total_energy", "param_0_real", "param_0_imag", "real_group_sampling_width"}
    assert all(len(v) == 3 for v in points.values())
    assert 100 < points["total_energy"][0] < 600
    means, errors = statistics.get_equilibrated_means(df, cutoff=400)
    assert errors == {} and abs(means["total_energy"]) < 0.2 and means["constant"] == 2.0
    auto_means, _ = statistics.get_equilibrated_means(df)
    assert abs(auto_means["constant"] - 2.0) < 1e-12
