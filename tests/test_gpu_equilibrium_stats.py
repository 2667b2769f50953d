"""GPU end-to-end test of save_equilibrium_stats (metropolis_engine.py:481-504) on a recorded single-chain run.
PARITY UNPINNED for the equilibration arithmetic (pymbar is absent): only the plumbing and sanity are checked."""
import numpy as np
import pytest

import metropolisengine_amd as me

pytestmark = pytest.mark.gpu


def test_save_equilibrium_stats_single_chain(capsys):
    # start far from the well so that the series has a visible relaxation
    eng = me.MetropolisEngine(me.DiagQuadratic((1.0, 4.0)), initial_real_params=[3.0, -2.0], temp=0.1, seed=9,
                              sampling_width=0.2)
    for _ in range(1500):
        eng.step_all(5)
        eng.measure()
    eng.save_equilibrium_stats()
    out = capsys.readouterr().out
    assert "global t_0" in out
    assert set(eng.eq_points) >= {"abs_param_0", "param_0_squared", "total_energy", "param_0", "param_1",
                                  "real_group_sampling_width"}
    assert 0 < eng.global_eq_point < 800
    assert eng.equilibrated_means["global_cutoff"] == eng.global_eq_point
    # after the cut-off the chain samples exp(-E/T): <x_i^2> = T/(2 a_i)
    assert abs(eng.equilibrated_means["param_0_squared"] - 0.05) < 0.03
    assert abs(eng.equilibrated_means["param_1_squared"] - 0.0125) < 0.01
    assert len(eng.df) == 1500
