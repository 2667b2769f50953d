"""The reference's mixed demo (demo/toymodel_complex_and_real.py): a uniform complex field c on a plane of size x * y,
E = k (1-x)^2 + k (1-y)^2 + x y (alpha |c|^2 + beta |c|^4), passed as the energy DICTIONARY
{"complex": {"field"}, "real": {"field", "area"}, "all": {"field", "area"}}, temp 0.1, 100 x (10 steps + 1 measure),
then the time series as a DataFrame -- on the GPU engine.

    python examples/demo_landau_field.py           (needs an MI355X and the built library)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metropolisengine_amd as me  # noqa: E402


def main(k=1.0, alpha=-1.0, beta=0.5, temp=0.1, n_measures=100, steps_per_measure=10, seed=2026):
    # terms=True: the two-term dictionary of the demo (the engine keeps one energy row per term)
    energy = me.LandauToy(k=k, alpha=alpha, beta=beta, terms=True)
    engine = me.MetropolisEngine(energy, initial_real_params=np.array([0.0, 0.0]),
                                 initial_complex_params=np.array([0 + 0j]), temp=temp, seed=seed)
    for _ in range(n_measures):
        for _ in range(steps_per_measure):
            engine.step_all()
        engine.measure()     # running means, covariance estimate, observables; one row of the time series
    print("plane size <x>, <y> = %s   field <c> = %s" % (engine.real_mean, engine.complex_mean))
    print("covariance of (x, y):\n%s\nvariance of c: %s" % (engine.covariance_matrix_real, engine.covariance_matrix_complex))
    for name, value in zip(engine.observables_names, engine.observables_mean):
        print("   <%s> = %.4f" % (name, value))
    print("energy terms", engine.energy)
    engine.save_time_series()          # prints the DataFrame, as the reference does
    return engine


if __name__ == "__main__":
    main()
