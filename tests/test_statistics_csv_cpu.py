"""The reference's time-series file format (exampledata300.csv, first 120 rows kept as the data fixture
tests/golden/reference_exampledata300_head.csv) read by statistics.timeseries_from_csv, and the host equilibration
statistics run on that real data."""
import os

import numpy as np

from metropolisengine_amd import statistics

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_exampledata300_head.csv")


def test_reference_csv_layout_is_parsed():
    series = statistics.timeseries_from_csv(FIXTURE)
    header = open(FIXTURE).readline().rstrip("\n").split(",")[1:]
    # 1 real + 9 complex parameters recorded by an engine with two energy terms (exampledata.csv)
    assert header[:10] == ["abs_param_%d" % i for i in range(10)]
    assert "field_energy" in header and "surface_energy" in header
    assert header[-1] == "complex_group_sampling_width" and "real_group_sampling_width" in header
    for name in header:
        assert (name in series) or (name + "_real" in series), name
    lengths = {len(v) for v in series.values()}
    assert lengths == {120}
    # complex parameters carry an imaginary part, observables (complex-typed with zero imaginary part) do not
    assert "param_5_imag" in series and "abs_param_5_imag" not in series
    first = open(FIXTURE).readlines()[1].rstrip("\n").split(",")
    col = 1 + header.index("param_5")
    assert series["param_5_real"][0] == complex(first[col]).real and series["param_5_imag"][0] == complex(first[col]).imag
    col = 1 + header.index("abs_param_5")
    assert series["abs_param_5_real"][0] == complex(first[col]).real
    mod = np.hypot(series["param_5_real"], series["param_5_imag"])
    assert np.allclose(mod, series["abs_param_5_real"], rtol=1e-12)            # the file is self-consistent
    single = statistics.timeseries_from_csv(FIXTURE, column_name="real_group_sampling_width")
    assert list(single) == ["real_group_sampling_width"]


def test_equilibration_statistics_run_on_reference_data():
    series = statistics.timeseries_from_csv(FIXTURE)
    for name in ("abs_param_0_real", "field_energy", "real_group_sampling_width"):
        t0, g, neff = statistics.detect_equilibration(series[name])
        assert 0 <= t0 < 119 and g >= 1.0 and 1.0 <= neff <= 121.0
    # the adaptive width is strongly autocorrelated, the parameter less so
    assert statistics.statistical_inefficiency(series["real_group_sampling_width"]) > 2.0


def test_frame_written_by_pandas_round_trips(tmp_path):
    """engine.df.to_csv(...) -> timeseries_from_csv: the complex parameter columns come back as _real / _imag."""
    import pandas
    rng = np.random.default_rng(2)
    frame = pandas.DataFrame({"abs_param_0": rng.random(30), "total_energy": rng.random(30),
                              "param_0": rng.random(30), "real_group_sampling_width": rng.random(30),
                              "param_1": rng.random(30) + 1j * rng.random(30),
                              "complex_group_sampling_width": rng.random(30)})
    path = tmp_path / "series.csv"
    frame.to_csv(path)
    back = statistics.timeseries_from_csv(str(path))
    assert set(back) == {"abs_param_0", "total_energy", "param_0", "real_group_sampling_width", "param_1_real",
                         "param_1_imag", "complex_group_sampling_width"}
    assert np.allclose(back["param_1_real"], frame["param_1"].to_numpy().real, rtol=1e-15)
    assert np.allclose(back["param_1_imag"], frame["param_1"].to_numpy().imag, rtol=1e-15)
    assert np.allclose(back["total_energy"], frame["total_energy"], rtol=1e-15)
