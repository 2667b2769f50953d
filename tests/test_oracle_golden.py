"""Pins oracle/reference_chain.py to the imported reference through the committed golden trajectories.

tests/golden/traj_*.npz were produced by oracle/make_golden.py from the reference itself
(metropolisengine/metropolis_engine.py) with injected random streams; this replays the same streams through
the restatement and requires agreement to 1e-12 (float64 round-off only) with identical accept decisions.
"""
import os
import random

import numpy as np
import pytest

from oracle import scenarios
from oracle.reference_chain import ReferenceChain, StreamSources, adaptation_constants

TOL = 1e-12


def replay(spec, gold):
    nr, nc = scenarios.dims(spec)
    src = StreamSources(gold["normals"], gold["uniforms"], nr, nc)
    chain = ReferenceChain(spec["energy"], initial_real_params=spec["real"], initial_complex_params=spec["cplx"],
                           temp=spec["temp"], reject_condition=spec.get("reject"), sources=src,
                           complex_sample_method=spec.get("method", "multivariate-gaussian"))
    return chain


@pytest.mark.parametrize("name", sorted(scenarios.SCENARIOS))
def test_trajectory_matches_reference(name, golden_dir):
    """Every scenario: the Gaussian step_all paths, group-wise stepping of mixed engines (step_real_group /
    step_complex_group called directly) and the magnitude-phase sampler."""
    spec = scenarios.SCENARIOS[name]
    gold = np.load(os.path.join(golden_dir, "traj_%s.npz" % name))
    nr, nc = scenarios.dims(spec)
    chain = replay(spec, gold)
    assert np.allclose([chain.alpha, chain.m, chain.ratio], gold["constants"], rtol=0, atol=1e-14)
    term_names = [str(t) for t in gold["term_names"]]
    stepper = {"all": chain.step_all, "real": chain.step_real_group, "complex": chain.step_complex_group}
    t = 0
    for k in range(spec["n_measures"]):
        for op in scenarios.ops(spec)[:-1]:
            accept = stepper[op]()
            assert (-1 if accept is None else int(accept)) == int(gold["accept"][t]), "decision differs at step %d" % t
            assert np.allclose(chain.real_params, gold["real_params"][t], rtol=0, atol=TOL)
            assert np.allclose(chain.complex_params, gold["complex_params"][t], rtol=0, atol=TOL)
            assert abs(chain.real_group_sampling_width - gold["real_width"][t]) < TOL
            assert abs(chain.complex_group_sampling_width - gold["complex_width"][t]) < TOL
            assert abs(np.real(chain.energy_total) - gold["energy_total"][t]) < TOL
            assert np.allclose([np.real(chain.energy[n]) for n in term_names], gold["energy_terms"][t],
                               rtol=0, atol=TOL)
            t += 1
        chain.measure()
        assert np.allclose(chain.real_mean, gold["real_mean"][k], rtol=0, atol=TOL)
        assert np.allclose(chain.complex_mean, gold["complex_mean"][k], rtol=0, atol=TOL)
        if nr:
            assert np.allclose(chain.covariance_matrix_real, gold["cov_real"][k], rtol=0, atol=TOL)
        if nc:
            assert np.allclose(chain.covariance_matrix_complex, gold["cov_complex"][k], rtol=0, atol=TOL)
        assert np.allclose(chain.observables_mean, gold["observables_mean"][k], rtol=0, atol=TOL)
    assert t == scenarios.n_steps(spec)


def test_seeded_legacy_anchor(golden_dir):
    """Unpatched legacy RNG (np.random.seed + random.seed): 1-D README config, 1000 x (step, measure).

    Values recorded from the imported reference (numpy 2.2.6); SURVEY.md section 4 quotes the seed-12345 row.
    """
    gold = np.load(os.path.join(golden_dir, "seeded_readme_1real.npz"))
    spec = scenarios.SCENARIOS["readme_1real"]
    for seed in gold.files:
        np.random.seed(int(seed))
        random.seed(int(seed))
        chain = ReferenceChain(spec["energy"], initial_real_params=spec["real"], temp=spec["temp"])
        for _ in range(1000):
            chain.step_all()
            chain.measure()
        got = np.array([chain.accepted, chain.real_mean[0], chain.covariance_matrix_real[0, 0],
                        chain.real_group_sampling_width, chain.observables_mean[0], chain.observables_mean[1],
                        chain.real_params[0]])
        assert np.allclose(got, gold[seed], rtol=0, atol=1e-12), seed
    assert np.allclose(gold["12345"][:4], [421, -0.00213108, 0.23508361, 0.517795253845864], atol=1e-8)


def test_adaptation_constants_survey_values():
    """ratio values listed in SURVEY.md A.1 (measured on the reference)."""
    want = {(1, 0): 4.7619047619, (2, 0): 3.4922480939, (4, 4): 2.5400055929, (2, 7): 2.5047373521,
            (16, 0): 2.3812985094, (64, 0): 2.2622681968}
    for (nr, nc), ratio in want.items():
        alpha, m, got = adaptation_constants(nr, nc, 0.3)
        assert abs(alpha - 1.0364333895) < 1e-9 and m == nr + nc
        assert abs(got - ratio) < 1e-9


def test_constructor_errors():
    with pytest.raises(ValueError):
        ReferenceChain(lambda r, c: 0.0)
    with pytest.raises(AssertionError):
        ReferenceChain(lambda r, c: 0.0, initial_real_params=[0.0], temp=-1)


@pytest.mark.skipif(not os.path.isdir("/root/reference/metropolisengine"), reason="the reference is only in the build container")
def test_fixtures_regenerate_bit_for_bit_from_the_reference():
    """oracle/make_golden.py --check: re-run the IMPORTED reference on the injected streams (PYTHONHASHSEED=0, which pins
    the order of its energy-term columns) and compare every array of every fixture with the committed file."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONHASHSEED="0", PYTHONDONTWRITEBYTECODE="1")
    res = subprocess.run([sys.executable, os.path.join(root, "oracle", "make_golden.py"), "--check"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-1000:]
    assert res.stdout.count("same ") == 13 and "DIFFERS" not in res.stdout
    # without the pinned hash seed the script refuses to run
    env.pop("PYTHONHASHSEED")
    res = subprocess.run([sys.executable, os.path.join(root, "oracle", "make_golden.py"), "--check"], env=env,
                         capture_output=True, text=True, timeout=60)
    assert res.returncode != 0 and "PYTHONHASHSEED=0" in res.stderr
