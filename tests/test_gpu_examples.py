"""The GPU counterparts of the reference's demo scripts (examples/) run end to end and land where the physics says."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
EXAMPLES = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples")


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(EXAMPLES, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_xy_potential_well_demo(capsys):
    single, many, stats = _load("demo_xy_well").main(n_measures=120, ensemble=1 << 14)
    assert single.measure_step_counter == 121 and single.covariance_matrix_real.shape == (2, 2)
    # stationary variance of exp(-c (x^2 + y^2) / T) is T / (2 c) = 0.05 per coordinate
    assert np.all(np.abs(np.diag(stats["covariance"]) - 0.05) < 6 * 0.05 * np.sqrt(2.0 / (1 << 14)))
    assert np.all(np.abs(stats["mean"]) < 6 * np.sqrt(0.05 / (1 << 14)))
    assert 0.2 < many.acceptance_rate() < 0.45
    assert "ensemble" in capsys.readouterr().out


def test_complex_and_real_demo(capsys):
    eng = _load("demo_landau_field").main(n_measures=60)
    assert set(eng.energy) == {"field", "area"}
    assert list(eng.df.columns) == ["abs_param_0", "abs_param_1", "abs_param_2", "param_0_squared", "param_1_squared",
                                    "area_energy", "field_energy", "param_0", "param_1", "real_group_sampling_width",
                                    "param_2", "complex_group_sampling_width"]
    assert len(eng.df) == 60
    x, y = eng.real_params
    a2 = abs(eng.complex_params[0]) ** 2
    assert abs(eng.energy["field"] - x * y * (-a2 + 0.5 * a2 * a2)) < 1e-5
    assert abs(eng.energy["area"] - ((1 - x) ** 2 + (1 - y) ** 2)) < 1e-5
    assert "energy terms" in capsys.readouterr().out


def test_readme_example_with_its_python_lambda(capsys):
    single, many = _load("demo_python_energy").main(n_steps=300, ensemble=1 << 14)
    assert single.measure_step_counter == 301 and np.shape(single.real_mean) == (1,)
    x = many.real_params[:, 0]
    assert abs(x.var() / 0.005 - 1.0) < 0.1 and abs(x.mean()) < 6 * np.sqrt(0.005 / (1 << 14))      # Var x = T / 2
    assert many.fused_cycles() == 30
    assert "ensemble of" in capsys.readouterr().out


def test_large_space_demo(capsys):
    """150 real parameters on the runtime-dimension set: the default per-chain adaptive shapes run (the reference's recursion,
    reproduced as it is), and one shape pooled over the ensemble brings the ensemble covariance to T/2 A^-1."""
    per_chain, pooled, err_ref, err_pool = _load("demo_large_space").main(n_chains=1 << 11, cycles=70, sweeps_per_cycle=20)
    assert per_chain.cov_mode == "reference" and per_chain.measure_step_counter == 71
    assert per_chain.covariance_matrix_real.shape == (1 << 11, 150, 150)
    assert 0.05 < per_chain.acceptance_rate() < 0.7 and err_ref < 1.0
    assert pooled.shared_factor() is not None and err_pool < 0.15
    assert "pooled over the ensemble" in capsys.readouterr().out
