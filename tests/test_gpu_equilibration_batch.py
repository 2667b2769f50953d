"""GPU test of the batched equilibration detection (me_detect_equilibration, csrc/me_statistics.hip) against the host
restatement in metropolisengine_amd/statistics.py (PARITY UNPINNED with respect to pymbar, see there)."""
import numpy as np
import pytest

import metropolisengine_amd as me
from metropolisengine_amd import statistics

pytestmark = pytest.mark.gpu


def _series(n_series, length, seed):
    rng = np.random.default_rng(seed)
    out = np.zeros((n_series, length))
    for s in range(n_series):
        phi = rng.uniform(0.0, 0.95)
        x = np.zeros(length)
        for i in range(1, length):
            x[i] = phi * x[i - 1] + rng.standard_normal()
        burn = int(rng.integers(0, length // 3))
        x[:burn] += np.linspace(rng.uniform(2.0, 8.0), 0.0, burn)       # a decaying transient of random length
        out[s] = x
    return out


@pytest.mark.parametrize("fast,nskip", [(True, 1), (False, 1), (True, 3)])
def test_batch_matches_host_restatement(fast, nskip):
    a = _series(24, 257, seed=3)
    a[5] = 1.25                                  # a constant series: (0, 1, 1) by convention
    t0, g, neff = statistics.detect_equilibration_batch(a, fast=fast, nskip=nskip)
    for s in range(a.shape[0]):
        want = statistics.detect_equilibration(a[s], fast=fast, nskip=nskip)
        assert t0[s] == want[0], (s, t0[s], want)
        assert abs(g[s] - want[1]) < 1e-9 * max(1.0, want[1])
        assert abs(neff[s] - want[2]) < 1e-9 * max(1.0, want[2])


def test_long_series_and_dataframe_path():
    import pandas
    a = _series(3, 1500, seed=9)
    t0, g, neff = statistics.detect_equilibration_batch(a)
    for s in range(3):
        want = statistics.detect_equilibration(a[s])
        assert t0[s] == want[0] and abs(g[s] - want[1]) < 1e-8 * want[1]
    df = pandas.DataFrame({"u": a[0], "v": a[1] + 1j * a[2], "const": np.ones(1500)})
    host = statistics.get_equilibration_points(df)
    dev = statistics.get_equilibration_points(df, device=0)
    assert set(host) == set(dev) == {"u", "v_real", "v_imag"}
    for key in host:
        assert host[key][0] == dev[key][0] and abs(host[key][1] - dev[key][1]) < 1e-8 * host[key][1]


def test_engine_equilibration_points_over_traced_chains():
    eng = me.MetropolisEngine(me.DiagQuadratic((1.0, 4.0)), initial_real_params=[3.0, -2.0], temp=0.1, n_chains=4096,
                              seed=5, trace_chains=16, trace_stride=256)
    for _ in range(200):
        eng.step_all(2)
        eng.measure()
    points = eng.equilibration_points()
    assert "param_0" in points and "total_energy" in points and "real_group_sampling_width" in points
    t0, g, neff = points["param_0"]
    assert t0.shape == (16,) and np.all(g >= 1.0) and np.all(neff >= 1.0)
    frame = eng.time_series_frame(3)
    want = statistics.detect_equilibration(frame["param_0"].to_numpy())
    assert t0[3] == want[0] and abs(g[3] - want[1]) < 1e-8 * want[1]
    assert np.median(t0) > 0            # the chains start far from equilibrium: a transient is detected


def test_batch_on_the_reference_data_file(golden_dir):
    """Every column of the reference's own recorded run (tests/golden/reference_exampledata300_head.csv)."""
    import os
    series = statistics.timeseries_from_csv(os.path.join(golden_dir, "reference_exampledata300_head.csv"))
    names = [k for k, v in series.items() if np.ptp(v) > 0]
    t0, g, neff = statistics.detect_equilibration_batch(np.stack([series[k] for k in names]))
    for i, name in enumerate(names):
        want = statistics.detect_equilibration(series[name])
        assert t0[i] == want[0] and abs(g[i] - want[1]) < 1e-8 * want[1], name
