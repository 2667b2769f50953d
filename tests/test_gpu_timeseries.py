"""GPU test of the time-series recording (metropolis_engine.py:31-35, :350-356) and the DataFrame layout of
save_time_series (:466-479): the reference's own frame, stored in the golden fixtures, against the frame the engine
builds from its device-side trace after replaying the same injected streams in float64."""
import os

import numpy as np
import pytest

import metropolisengine_amd as me
from metropolisengine_amd import _capi
from oracle import scenarios

pytestmark = pytest.mark.gpu
TOL = 1e-9

SPECS = {
    "readme_1real": (me.IsoQuadratic(1.0), "multivariate-gaussian"),
    "aniso_2complex": (me.DiagQuadratic((), (1.0, 3.0)), "multivariate-gaussian"),
    "landau_toy": (me.LandauToy(1.0, -1.0, 0.5), "multivariate-gaussian"),
    "magphase_1real_2complex": (me.DiagQuadratic((0.5,), (1.0, 3.0)), "magnitude-phase"),
}


@pytest.mark.parametrize("name", sorted(SPECS))
def test_time_series_frame_matches_the_reference(name, golden_dir, capsys):
    spec = scenarios.SCENARIOS[name]
    gold = np.load(os.path.join(golden_dir, "traj_%s.npz" % name))
    energy_spec, method = SPECS[name]
    nr, nc = scenarios.dims(spec)
    eng = me.MetropolisEngine(energy_spec, None, spec["real"], spec["cplx"], temp=spec["temp"], n_chains=1, dtype="f64",
                              complex_sample_method=method)
    assert eng.trace_chains == 1                     # a single-chain engine records like the reference does
    t = 0
    for _ in range(spec["n_measures"]):
        for op in scenarios.ops(spec)[:-1]:
            if op == "complex" and method == "magnitude-phase":
                eng.step_injected(gold["normals"][t:t + 1, None, :nc], gold["uniforms"][t:t + 1, None, :],
                                  kind=_capi.STEP_COMPLEX_MAGNITUDE_PHASE)
            else:
                kind = {"all": _capi.STEP_ALL, "real": _capi.STEP_REAL_GROUP, "complex": _capi.STEP_COMPLEX_GROUP}[op]
                eng.step_injected(gold["normals"][t:t + 1, None, :], gold["uniforms"][t:t + 1, :1], kind=kind)
            t += 1
        eng.measure()
    eng.save_time_series()
    assert "abs_param_0" in capsys.readouterr().out   # the reference prints the frame (:479)
    frame = eng.df
    ref_cols = [str(c) for c in gold["df_columns"]]
    # same columns in the same order, except that the reference has one "<term>_energy" column per energy term
    # (stale in mixed engines, quirk Q5) where this engine has the current "total_energy"
    ours = [c for c in frame.columns if not c.endswith("_energy")]
    theirs = [c for c in ref_cols if not c.endswith("_energy")]
    assert ours == theirs
    assert [c for c in frame.columns if c.endswith("_energy")] == ["total_energy"]
    first_energy = min(i for i, c in enumerate(ref_cols) if c.endswith("_energy"))
    assert list(frame.columns).index("total_energy") == first_energy
    assert len(frame) == spec["n_measures"]
    for col in ours:
        want = gold["df_values"][:, ref_cols.index(col)]
        got = np.asarray(frame[col].to_numpy(), dtype=np.complex128)
        assert np.allclose(got, want, rtol=0, atol=TOL), col
    # the list attributes the reference exposes
    if nr:
        assert len(eng.real_params_time_series) == spec["n_measures"]
        assert np.allclose(eng.real_params_time_series[-1], gold["real_params"][-1], atol=TOL)
    else:
        assert eng.real_params_time_series is None and eng.real_group_sampling_width_time_series is None
    if nc:
        assert np.allclose(eng.complex_params_time_series[-1], gold["complex_params"][-1], atol=TOL)
    else:
        assert eng.complex_params_time_series is None
    assert len(eng.observables_time_series) == spec["n_measures"]
    assert abs(eng.energy_time_series["total"][-1] - eng.energy_total) < 1e-12


def test_strided_traces_of_a_many_chain_engine():
    n = 4096
    eng = me.MetropolisEngine(me.DiagQuadratic((1.0, 2.0), (1.5, 3.0)), None, [0.1, -0.1], [0.2 + 0.1j, -0.1 + 0.3j],
                              temp=1.0, n_chains=n, seed=5, trace_chains=8, trace_stride=500)
    rows = 1100                                      # crosses the 1024-row growth of the device-side series
    snapshots = {}
    for k in range(rows):
        eng.step_all(2) if k % 3 else eng.step_real_group(2)
        eng.measure()
        if k in (0, 700, rows - 1):
            snapshots[k] = (eng._get(0), eng._get(1), eng._get(2))
    tr = eng.trace()
    assert tr.shape == (rows, 8, 6 + 1 + 3)
    for k, (x, energy, width) in snapshots.items():
        chains = np.arange(8) * 500
        assert np.array_equal(tr[k, :, :6].astype(np.float32), x[chains].astype(np.float32))
        assert np.array_equal(tr[k, :, 6].astype(np.float32), energy[chains, 0].astype(np.float32))
        assert np.array_equal(tr[k, :, 7:].astype(np.float32), width[chains].astype(np.float32))
    frames = [eng.time_series_frame(c) for c in range(8)]
    assert all(len(f) == rows for f in frames)
    assert list(frames[0].columns) == ["abs_param_0", "abs_param_1", "abs_param_2", "abs_param_3", "param_0_squared",
                                       "param_1_squared", "total_energy", "param_0", "param_1",
                                       "real_group_sampling_width", "param_2", "param_3", "complex_group_sampling_width"]
    with pytest.raises(ValueError):
        eng.time_series_frame(8)
    quiet = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 16, None, temp=1.0, n_chains=n)
    quiet.measure()
    assert quiet.trace().shape[:2] == (0, 0)         # many-chain engines do not record unless asked
