"""GPU test of a user-written hard-wall predicate (ME_REJECT_USER): the example plugin's me_user_reject, |x0| >= 1,
must act exactly like the built-in AbsReal0AtLeast(1.0) on the same streams (device-side form of the reference's
reject_condition callback, metropolis_engine.py:142-146, :247-249)."""
import os

import numpy as np
import pytest

import metropolisengine_amd as me
from oracle import energies
from oracle.manychain import ManyChainOracle

pytestmark = pytest.mark.gpu
SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "user_energy_cylinder.h")
COEF = (1.0, 0.5, 1.0)


def test_user_reject_equals_builtin_wall():
    real0, cplx0 = [0.9, 0.0], [0.05] * 7          # start next to the wall so that it is hit constantly
    kw = dict(temp=2.0, n_chains=4096, seed=8, sampling_width=0.2)
    builtin = me.MetropolisEngine(me.UserEnergy("cylinder", SRC, COEF), me.AbsReal0AtLeast(1.0), real0, cplx0, **kw)
    user = me.MetropolisEngine(me.UserEnergy("cylinder", SRC, COEF), me.UserReject(), real0, cplx0, **kw)
    free = me.MetropolisEngine(me.UserEnergy("cylinder", SRC, COEF), None, real0, cplx0, **kw)
    for eng in (builtin, user, free):
        for _ in range(20):
            eng.step_all(5)
            eng.measure()
    for field in range(7):
        assert np.array_equal(builtin._get(field), user._get(field)), field
    assert builtin.accept_stats() == user.accept_stats()
    assert np.all(np.abs(user._get(0)[:, 0]) < 1.0)
    assert np.any(np.abs(free._get(0)[:, 0]) >= 1.0)    # without the wall chains do leave (-1, 1)


def test_user_reject_f64_follows_the_oracle():
    real0, cplx0 = [0.9, 0.0], [0.05] * 7
    eng = me.MetropolisEngine(me.UserEnergy("cylinder", SRC, COEF, indirect=True), me.UserReject(), real0, cplx0,
                              temp=2.0, n_chains=128, seed=8, sampling_width=0.2, dtype="f64")
    ora = ManyChainOracle(2, 7, energies.cylinder_surrogate(2, 7, *COEF), 128, seed=8, temp=2.0,
                          initial_real_params=real0, initial_complex_params=cplx0, sampling_width=0.2,
                          reject=energies.wall_reject(1.0))
    for _ in range(30):
        eng.step_all(4)
        ora.step(4)
    assert np.allclose(eng._get(0), ora.x, rtol=0, atol=1e-9)
    assert eng.accept_stats() == (ora.accepted, ora.proposed)


def test_user_reject_needs_a_user_energy():
    with pytest.raises(ValueError):
        me.MetropolisEngine(me.CylinderSurrogate(*COEF), me.UserReject(), [0.1, 0.0], [0.05] * 7, temp=0.1, n_chains=64)
